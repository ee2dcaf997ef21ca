#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_rules.py -m gpu -q -x -p no:cacheprovider -k "linear or golden or module" 2>&1 | tail -3 )
( timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 )
