#!/bin/bash
# GELU producers that emit operand planes: parity (bitwise vs the separate split passes), model-level tests, bench A/B
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "gelu or mlp_block or linear_layer or reuses_forward or fused_producers or bert_base_with_layer" 2>&1 | tail -15 ) > gpurun_out/t38_tests.log
for f in 0 1 0 1; do
  ( echo -n "TE_X6_FUSE_GELU=$f "; TE_X6_FUSE_GELU=$f timeout 300 python bench.py --steps 10 --cpu-baseline off --no-roofline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" ) >> gpurun_out/t38_ab.log
done
cat gpurun_out/t38_tests.log gpurun_out/t38_ab.log
