#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum" 2>&1 | tail -3 ) > gpurun_out/t8_tests.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  ( timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -2 ) >> gpurun_out/t8_bench.log
done
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t8_build_study.log
( timeout 120 python scripts/attn_kb_prof.py 64 12 197; timeout 120 python scripts/attn_kb_prof.py 32 12 512 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/t8_prof.log
cat gpurun_out/t8_tests.log gpurun_out/t8_bench.log gpurun_out/t8_prof.log
