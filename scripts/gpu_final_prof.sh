#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof
( timeout 300 python bench.py > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err ); tail -8 gpurun_out/bench_b64.err
cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /root/repo/gpurun_out/prof -o bench -- \
    python /root/repo/bench.py --cpu-baseline off > /root/repo/gpurun_out/prof_bench.json 2> /root/repo/gpurun_out/prof_bench.err
cd /root/repo; rm -f gpurun_out/prof/*agent_info*; find gpurun_out/prof -name '*kernel_trace.csv' -size +30M -delete
cut -c1-200 gpurun_out/prof_bench.json; tail -3 gpurun_out/prof_bench.err; ls -la gpurun_out/prof
