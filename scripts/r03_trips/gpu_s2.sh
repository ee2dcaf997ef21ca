#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python benchmarks/x6_study.py --iters 8 > gpurun_out/s2_x6_study.log 2>&1
echo "study rc=$?" >> gpurun_out/s2_x6_study.log
cat gpurun_out/s2_x6_study.log | grep -v amdgpu.ids
bash scripts/x6_pmc.sh > gpurun_out/s2_x6_pmc.log 2>&1
tail -120 gpurun_out/s2_x6_pmc.log
