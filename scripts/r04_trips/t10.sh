#!/bin/bash
# trip 10: model-level tests (lrp variant on x6, batch-64 parity with 19 oracle samples, configs 2 / 3 with four each)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "orig_lrp or config1 or config2 or config3 or zz_band or linear_x6_path" > gpurun_out/t10_tests.log 2>&1
grep -v amdgpu gpurun_out/t10_tests.log | tail -25
