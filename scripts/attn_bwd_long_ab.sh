#!/bin/bash
# same-box A/B of the long-sequence attention backward: row side on te_attn_bwd6l.hip (default with the forward output at hand) vs
# attn_bwd_rows_kernel of te_attn_long.hip (TE_ATTN_BWD_LONG=old, measurement build TE_BUILD_DEFINES=TE_STUDY); the producer tests
# run on the shipped library first.  attn_bench.py hands `out` to the backward.
L=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
timeout 900 python -m pytest tests/test_gpu_producers.py -q -m gpu -x -k "attention" 2>&1 | tail -15
for shape in "32 16 577" "32 12 512" "16 12 640" "32 12 300"; do
  for v in new old w4 new old w4; do
    unset TE_ATTN_BWD_LONG TE_BWD6L_WAVES
    if [ $v = old ]; then export TE_ATTN_BWD_LONG=old; fi
    if [ $v = w4 ]; then export TE_BWD6L_WAVES=4; fi
    echo -n "$v  "; TE_RELPROP_LIB=$L timeout 300 python scripts/attn_bench.py $shape 64 producers 2>&1 | grep "producer forward"
  done
done
