#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t36_build_study.log
( TE_ATTN_AV_SCHED=1 timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum" 2>&1 | tail -4 ) > gpurun_out/t36_tests.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  for sch in 0 1 0 1; do
    ( echo -n "sched=$sch "; TE_ATTN_AV_SCHED=$sch timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' '; echo ) >> gpurun_out/t36_ab.log
  done
done
( TE_ATTN_AV_SCHED=1 timeout 120 python scripts/attn_determinism.py rules 2>&1 | grep -v amdgpu | head -3 ) >> gpurun_out/t36_ab.log
cat gpurun_out/t36_tests.log gpurun_out/t36_ab.log
