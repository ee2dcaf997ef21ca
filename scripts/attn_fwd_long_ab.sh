#!/bin/bash
# same-box A/B of the long-sequence attention forward: te_attn_fwd6l.hip (default: 4-wave workgroups, two per CU; w8: TE_FWD6L_WAVES=8,
# one 8-wave workgroup per CU) vs the round-3 kernel of te_attn_long.hip (TE_ATTN_FWD_LONG=old); measurement build
# TE_BUILD_DEFINES=TE_STUDY; the producer tests run on the shipped library first
L=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
timeout 900 python -m pytest tests/test_gpu_producers.py -q -m gpu -x -k "attention_forward_producer or attention_producer_bert" 2>&1 | tail -15
for shape in "32 16 577" "32 12 512" "16 12 640" "32 12 300"; do
  for v in new w8 old new w8 old; do
    unset TE_ATTN_FWD_LONG TE_FWD6L_WAVES
    if [ $v = old ]; then export TE_ATTN_FWD_LONG=old; fi
    if [ $v = w8 ]; then export TE_FWD6L_WAVES=8; fi
    echo -n "$v  "; TE_RELPROP_LIB=$L timeout 300 python scripts/attn_bench.py $shape 64 producers 2>&1 | grep "producer forward"
  done
done
