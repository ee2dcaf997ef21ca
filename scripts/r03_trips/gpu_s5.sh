#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
sed -i 's/NAMES = {0:/NAMES = {5: "shipped+stamps", 6: "no-epilogue+stamps", 0:/' benchmarks/x6_study.py
timeout 300 python benchmarks/x6_study.py --iters 8 --studies 0,5,4,6 > gpurun_out/s5_x6_study.log 2>&1
grep -v amdgpu.ids gpurun_out/s5_x6_study.log | cut -c1-400
