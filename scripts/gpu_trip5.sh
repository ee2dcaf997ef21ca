#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for bn in 128 64; do
for a in 0 1 2 3; do
  L=""; [ $a != 0 ] && L="--lib benchmarks/libte_ablate$a.so"
  ( TE_LINEAR_BN=$bn timeout 120 python benchmarks/linear_bench.py --skip-peak $L 2>&1 | grep -v amdgpu.ids | head -9 ) > gpurun_out/abl_${bn}_$a.log
  echo "=== BN=$bn ablation $a ==="; cat gpurun_out/abl_${bn}_$a.log
done; done
