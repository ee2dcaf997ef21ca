#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python benchmarks/x6_graph_debug.py > gpurun_out/s8_graph_debug.log 2>&1
grep -v amdgpu.ids gpurun_out/s8_graph_debug.log | tail -20
( timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "batch_equals_singles or config1" 2>&1 | grep -E "^E  |Error|assert|passed|failed" | cut -c1-300 | head -40 ) > gpurun_out/s8_tests.log
cat gpurun_out/s8_tests.log
