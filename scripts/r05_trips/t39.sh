#!/bin/bash
# GELU plane producers: direct mapping vs LDS-staged coalesced mapping, per kernel; parity of the staged one; bench A/B
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
STUDY=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
( TE_RELPROP_LIB=$STUDY TE_GELU_SPLIT=direct timeout 120 python scripts/gelu_planes_bench.py 2>&1 | grep -v amdgpu ) > gpurun_out/t39_kernels.log
( TE_RELPROP_LIB=$STUDY timeout 120 python scripts/gelu_planes_bench.py 2>&1 | grep -v amdgpu ) >> gpurun_out/t39_kernels.log
( timeout 600 python -m pytest tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "gelu or mlp_block" 2>&1 | tail -4 ) > gpurun_out/t39_tests.log
for f in 0 1 0 1; do
  ( echo -n "TE_X6_FUSE_GELU=$f "; TE_X6_FUSE_GELU=$f timeout 300 python bench.py --steps 10 --cpu-baseline off --no-roofline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" ) >> gpurun_out/t39_ab.log
done
cat gpurun_out/t39_kernels.log gpurun_out/t39_tests.log gpurun_out/t39_ab.log
