#!/bin/bash
# Tuning study of the one-product Z-pass kernel (linear_k1_kernel<ZM_FWD>): TE_ZFWD_VARIANT x tile pin, per shape.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for v in 0 1 2 3; do
  for t in auto 128x128 64x64; do
    if [ "$t" = auto ]; then unset TE_LINEAR_TILE; else export TE_LINEAR_TILE=$t; fi
    echo "== TE_ZFWD_VARIANT=$v TE_LINEAR_TILE=$t"
    TE_ZFWD_VARIANT=$v python benchmarks/linear_bench.py --skip-peak --reps 5 2>&1 | grep -E "zfwd|cpass"
  done
done 2>&1 | tee gpurun_out/zfwd_variants.log
# correctness of the candidate variants on the rule tests
for v in 1 2; do
  echo "== pytest TE_ZFWD_VARIANT=$v"
  TE_ZFWD_VARIANT=$v python -m pytest tests/test_gpu_rules.py -q -m gpu -k "linear" 2>&1 | tail -3
done 2>&1 | tee -a gpurun_out/zfwd_variants.log
