#!/bin/bash
# same-box A/B of the attention forward producer: te_attn_fwd6.hip (default) vs the round-2 kernel (TE_ATTN_FWD=old, measurement build)
L=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
timeout 600 python -m pytest tests/test_gpu_producers.py -q -m gpu -x -k "attention_forward" 2>&1 | tail -3
for v in new old new old; do
  if [ $v = old ]; then export TE_ATTN_FWD=old; else unset TE_ATTN_FWD; fi
  echo -n "$v  "; TE_RELPROP_LIB=$L python scripts/attn_bench.py 64 12 197 64 producers 2>&1 | grep producer
done
for shape in "32 12 128" "64 12 224" "64 12 160"; do
  for v in new old; do
    if [ $v = old ]; then export TE_ATTN_FWD=old; else unset TE_ATTN_FWD; fi
    echo -n "$v  "; TE_RELPROP_LIB=$L python scripts/attn_bench.py $shape 64 producers 2>&1 | grep producer
  done
done
