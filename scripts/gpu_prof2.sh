#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/prof_full -o bench -- \
    python /root/repo/bench.py --steps 2 --warmup 1 --cpu-baseline off --no-roofline > /root/repo/gpurun_out/prof_full.json 2> /root/repo/gpurun_out/prof_full.err
cd /root/repo; rm -f gpurun_out/prof_full/*agent_info* gpurun_out/prof_full/*kernel_trace*
