#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t26_build_study.log
( timeout 120 python scripts/attn_kb_prof.py 64 12 197 qk; timeout 120 python scripts/attn_kb_prof.py 32 12 512 qk ) 2>&1 | grep -v amdgpu.ids > gpurun_out/t26_prof.log
cat gpurun_out/t26_prof.log
