#!/bin/bash
# QK x6 kb kernel after the in-flight-register fix: parity, determinism in the replayed step (30 replays), time vs the round-2 kernel
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
export TE_RELPROP_LIB=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
( TE_ATTN_QK=x6 timeout 400 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum" 2>&1 | tail -4 ) > gpurun_out/t37_tests.log
( TE_ATTN_QK=x6 timeout 120 python scripts/attn_qk_check.py 2>&1 | grep -v amdgpu | tail -12 ) > gpurun_out/t37_qk_check.log
( echo "== AV=x6 QK=x6, 30 graph replays vs serial eager:"; TE_ATTN_QK=x6 timeout 400 python scripts/graph_vs_eager.py 64 30 2>&1 | grep -v amdgpu > gpurun_out/t37_gve_full.log; grep -c "graph replay.*bitwise equal" gpurun_out/t37_gve_full.log; grep -c DIFFERENT gpurun_out/t37_gve_full.log; grep DIFFERENT gpurun_out/t37_gve_full.log | head -5 ) > gpurun_out/t37_gve.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  for qk in old x6 old x6; do
    ( echo -n "QK=$qk "; TE_ATTN_QK=$qk timeout 120 python scripts/attn_bench.py $shape 2>&1 | grep -v amdgpu.ids | tail -1 ) >> gpurun_out/t37_ab.log
  done
done
cat gpurun_out/t37_tests.log gpurun_out/t37_qk_check.log gpurun_out/t37_gve.log gpurun_out/t37_ab.log
