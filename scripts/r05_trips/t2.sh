#!/bin/bash
# trip 2: AV kb kernel with the tile's other instructions placed between the row product's MFMAs
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum or vit or bert" 2>&1 | tail -8 ) > gpurun_out/t2_tests.log
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t2_build_study.log
for shape in "64 12 197 64" "32 16 577 64" "32 12 512 64"; do
  for impl in old new old new; do
    ( echo "impl=$impl"; TE_ATTN_AV=$impl timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -3 ) >> gpurun_out/t2_attn_ab.log
  done
done
echo "=== tests ==="; cat gpurun_out/t2_tests.log
echo "=== A/B ==="; cat gpurun_out/t2_attn_ab.log
