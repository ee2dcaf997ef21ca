#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum" 2>&1 | tail -12 ) > gpurun_out/t18_tests.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  ( timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -2 ) >> gpurun_out/t18_bench.log
done
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t18_build_study.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  for impl in old fp32kb x6 old fp32kb x6; do
    ( echo -n "impl=$impl "; TE_ATTN_AV=$impl timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' '; echo ) >> gpurun_out/t18_ab.log
  done
done
cat gpurun_out/t18_tests.log gpurun_out/t18_bench.log gpurun_out/t18_ab.log
