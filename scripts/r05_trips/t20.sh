#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum" 2>&1 | tail -15 ) > gpurun_out/t20_tests.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  ( timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -2 ) >> gpurun_out/t20_bench.log
done
cat gpurun_out/t20_tests.log gpurun_out/t20_bench.log
