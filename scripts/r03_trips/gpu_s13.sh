#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
rm -rf gpurun_out/prof_vitl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d "$ROOT/gpurun_out/prof_vitl" -o bench -- \
    python "$ROOT/bench.py" --config vit_l16_384 --steps 2 --warmup 1 --cpu-baseline off --no-roofline > /dev/null 2> "$ROOT/gpurun_out/prof_vitl.err" )
rm -f gpurun_out/prof_vitl/*agent_info* gpurun_out/prof_vitl/*kernel_trace*
head -30 gpurun_out/prof_vitl/bench_kernel_stats.csv | cut -d, -f1-5
