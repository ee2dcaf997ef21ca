#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum" 2>&1 | tail -3 ) > gpurun_out/t11_tests.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  ( timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -2 ) >> gpurun_out/t11_bench.log
done
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t11_build_study.log
for shape in "64 12 197 64" "32 12 512 64"; do
  for st in 0 1 2 9 8 0; do
    ( echo -n "study=$st "; TE_ATTN_KB_STUDY=$st timeout 120 python scripts/attn_bench.py $shape 2>&1 | grep -v amdgpu.ids | tail -1 ) >> gpurun_out/t11_av_study.log
  done
done
cat gpurun_out/t11_tests.log gpurun_out/t11_bench.log gpurun_out/t11_av_study.log
