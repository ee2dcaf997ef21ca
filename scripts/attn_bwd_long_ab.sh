#!/bin/bash
# same-box A/B of the long-sequence attention backward (te_attn_bwd6l.hip; attn_bench.py hands `out` to the backward):
#   new      row side bwd6l_rows_kernel + column side bwd6l_cols_kernel (default)
#   oldcols  new row side, attn_bwd_cols_kernel of te_attn_long.hip (TE_ATTN_BWD_COLS=old)
#   old      both round-3 kernels (TE_ATTN_BWD_LONG=old TE_ATTN_BWD_COLS=old)
#   w4       new kernels, 4-wave workgroups forced (TE_BWD6L_WAVES=4)
# measurement build TE_BUILD_DEFINES=TE_STUDY; the producer tests run on the shipped library first
L=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
timeout 900 python -m pytest tests/test_gpu_producers.py -q -m gpu -x -k "attention" 2>&1 | tail -15
for shape in "32 16 577" "32 12 512" "16 12 640" "32 12 300"; do
  for v in new oldcols old w4 new oldcols old w4; do
    unset TE_ATTN_BWD_LONG TE_BWD6L_WAVES TE_ATTN_BWD_COLS
    if [ $v = old ]; then export TE_ATTN_BWD_LONG=old TE_ATTN_BWD_COLS=old; fi
    if [ $v = oldcols ]; then export TE_ATTN_BWD_COLS=old; fi
    if [ $v = w4 ]; then export TE_BWD6L_WAVES=4; fi
    echo -n "$v  "; TE_RELPROP_LIB=$L timeout 300 python scripts/attn_bench.py $shape 64 producers 2>&1 | grep "producer forward"
  done
done
