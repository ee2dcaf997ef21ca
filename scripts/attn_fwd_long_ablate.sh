#!/bin/bash
# what the long-sequence forward's time is made of: measurement switches of te_attn_fwd6l.hip (TE_FWD6L_OPT, study build), same box
L=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
for shape in "32 12 512" "32 16 577"; do
  for w in 4 8; do
    for opt in 0 1 2 4 8 16 31; do
      echo -n "W=$w OPT=$opt  "; TE_FWD6L_WAVES=$w TE_FWD6L_OPT=$opt TE_RELPROP_LIB=$L timeout 300 python scripts/attn_bench.py $shape 64 producers 2>&1 | grep "producer forward"
    done
  done
done
