#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_rules.py -m gpu -q -x -p no:cacheprovider -k "linear or golden or module" 2>&1 | tail -3 )
for st in 0 100 50 200; do
echo "=== auto BN persistent stagger $st ==="; ( TE_LINEAR_STAGGER=$st timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 )
done
echo "=== auto BN NON-persistent stagger 100 ==="; ( TE_LINEAR_PERSIST=0 TE_LINEAR_STAGGER=100 timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 )
