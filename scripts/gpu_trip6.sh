#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for pr in 0 1; do
  ( TE_LINEAR_PRIO=$pr timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 ) > gpurun_out/prio_$pr.log
  echo "=== auto BN, prio $pr ==="; cat gpurun_out/prio_$pr.log
done
( timeout 300 python -m pytest tests/test_gpu_rules.py -m gpu -q -x -p no:cacheprovider -k "linear or golden" 2>&1 | tail -5 )
