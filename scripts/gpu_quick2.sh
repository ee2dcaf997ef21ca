#!/bin/bash
export TMPDIR=/tmp
for t in 128x128 128x64 64x64; do
echo "=== tile $t ==="; ( TE_LINEAR_TILE=$t timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep "zfwd" )
done
