#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python benchmarks/x6_prof.py --tile 0 > gpurun_out/s4_x6_prof.log 2>&1
timeout 300 python benchmarks/x6_prof.py --tile 1 >> gpurun_out/s4_x6_prof.log 2>&1
grep -v amdgpu.ids gpurun_out/s4_x6_prof.log | cut -c1-700
