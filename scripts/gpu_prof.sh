#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /root/repo/gpurun_out/prof -o bench -- \
    python /root/repo/bench.py --steps 3 --warmup 1 --cpu-baseline off > /root/repo/gpurun_out/prof_bench.json 2> /root/repo/gpurun_out/prof_bench.err
cd /root/repo; rm -f gpurun_out/prof/*agent_info*
cut -c1-300 gpurun_out/prof_bench.json
