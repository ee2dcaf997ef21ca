#!/bin/bash
# trip 4: ablations of the AV kb kernel (study build): which part of the tile costs what
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t4_build_study.log
for st in 0 1 2 3 4 5 6 0; do
  ( echo "study=$st"; TE_ATTN_KB_STUDY=$st timeout 120 python scripts/attn_bench.py 64 12 197 64 2>&1 | grep -v amdgpu.ids | tail -1 ) >> gpurun_out/t4_av_study.log
done
for st in 0 1 2 3 4 5; do
  ( echo "study=$st"; TE_ATTN_KB_STUDY=$st timeout 120 python scripts/attn_bench.py 32 12 512 64 2>&1 | grep -v amdgpu.ids | tail -1 ) >> gpurun_out/t4_av_study.log
done
cat gpurun_out/t4_av_study.log
