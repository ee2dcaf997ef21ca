#!/bin/bash
# same-box A/B of the long-sequence QK rule: te_attn_qk6l.hip (default) vs qk_rule_kernel + qk_finish_kernel of te_attn_rules.hip
# (TE_ATTN_QK_LONG=old, measurement build TE_BUILD_DEFINES=TE_STUDY); the rule tests run on the shipped library first
L=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
timeout 900 python -m pytest tests/test_gpu_rules.py -q -m gpu -x -k "attention_rules or einsum or matmul" 2>&1 | tail -6
for shape in "32 16 577" "32 12 512" "16 12 640" "32 12 300"; do
  for v in new old w4 new old w4; do
    unset TE_ATTN_QK_LONG TE_QK6L_WAVES
    if [ $v = old ]; then export TE_ATTN_QK_LONG=old; fi
    if [ $v = w4 ]; then export TE_QK6L_WAVES=4; fi
    echo -n "$v  "; TE_RELPROP_LIB=$L timeout 300 python scripts/attn_bench.py $shape 64 2>&1 | grep "QK rule"
  done
done
