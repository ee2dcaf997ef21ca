#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t10_build_study.log
for shape in "64 12 197 64" "32 12 512 64"; do
  for st in 0 1 2 9 4 5 8 0; do
    ( echo -n "study=$st "; TE_ATTN_KB_STUDY=$st timeout 120 python scripts/attn_bench.py $shape 2>&1 | grep -v amdgpu.ids | tail -1 ) >> gpurun_out/t10_av_study.log
  done
done
cat gpurun_out/t10_av_study.log
