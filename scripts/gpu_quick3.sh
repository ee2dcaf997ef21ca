#!/bin/bash
export TMPDIR=/tmp
for v in 0 1; do
echo "=== TORCH_BLAS_PREFER_HIPBLASLT=$v ==="
( TORCH_BLAS_PREFER_HIPBLASLT=$v timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline off 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|timed.*' )
done
