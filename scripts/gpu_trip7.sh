#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_rules.py -m gpu -q -x -p no:cacheprovider -k "linear or golden or module" 2>&1 | tail -5 )
for pe in 1 0; do
  ( TE_LINEAR_PERSIST=$pe timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 ) > gpurun_out/persist_$pe.log
  echo "=== auto BN, persistent $pe ==="; cat gpurun_out/persist_$pe.log
done
for bn in 128 64; do
  ( TE_LINEAR_BN=$bn timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 ) > gpurun_out/persist_bn$bn.log
  echo "=== BN $bn, persistent ==="; cat gpurun_out/persist_bn$bn.log
done
