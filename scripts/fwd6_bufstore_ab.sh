#!/bin/bash
# same-box A/B of fwd6's block stores: buffer stores for the inner key blocks (shipped library) vs guarded global stores for every
# block (study build: TE_BUILD_DEFINES=TE_FWD6_NO_BUFSTORE), plus the producer tests on the shipped library
S=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
timeout 900 python -m pytest tests/test_gpu_producers.py -q -m gpu -x -k "attention_forward or attention_producer" 2>&1 | tail -3
for rep in 1 2; do
  for shape in "64 12 197" "64 12 224" "64 12 160" "32 12 128"; do
    echo -n "buf  $shape  "; python scripts/attn_bench.py $shape 64 producers 2>&1 | grep "producer forward"
    echo -n "glob $shape  "; TE_RELPROP_LIB=$S python scripts/attn_bench.py $shape 64 producers 2>&1 | grep "producer forward"
  done
done
