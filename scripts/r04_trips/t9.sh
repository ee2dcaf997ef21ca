#!/bin/bash
# trip 9: lrp / alpha != 1 on the x6 kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rules.py -x -q -m gpu -k "x6 or linear" > gpurun_out/t9_tests.log 2>&1
grep -v amdgpu gpurun_out/t9_tests.log | tail -25
