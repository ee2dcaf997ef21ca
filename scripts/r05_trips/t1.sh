#!/bin/bash
# trip 1: new AV kernels (te_attn_kb.hip) -- parity tests of the attention rules / producers, RCCL one-rank test, x6 flag
# refusal test; microbenchmarks default build; then a TE_STUDY build for the same-box A/B old vs new AV kernel.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py tests/test_gpu_parallel.py -m gpu -q -p no:cacheprovider -x --durations=5 2>&1 | tail -25 ) > gpurun_out/t1_tests.log
for shape in "64 12 197 64" "32 16 577 64" "32 12 512 64"; do
  ( timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | tail -3 ) >> gpurun_out/t1_attn_new.log
done
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t1_build_study.log
for shape in "64 12 197 64" "32 16 577 64" "32 12 512 64"; do
  for impl in old new old new; do
    ( echo "impl=$impl"; TE_ATTN_AV=$impl timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | tail -3 ) >> gpurun_out/t1_attn_ab.log
  done
done
echo "=== tests ==="; cat gpurun_out/t1_tests.log
echo "=== new ==="; cat gpurun_out/t1_attn_new.log
echo "=== A/B ==="; cat gpurun_out/t1_attn_ab.log
