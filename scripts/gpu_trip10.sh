#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_rules.py -m gpu -q -x -p no:cacheprovider -k "linear or golden or module" 2>&1 | tail -3 )
for t in 64x64 128x64 128x128 auto; do
echo "=== tile $t ==="; ( TE_LINEAR_TILE=$t timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 )
done
echo "=== tile 64x64 persistent ==="; ( TE_LINEAR_PERSIST=1 TE_LINEAR_TILE=64x64 timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 )
