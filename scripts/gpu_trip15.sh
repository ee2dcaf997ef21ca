#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/tests_full.log
echo "=== tests ==="; cat gpurun_out/tests_full.log
cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /root/repo/gpurun_out/prof -o bench -- \
    python /root/repo/bench.py --cpu-baseline off > /root/repo/gpurun_out/prof_bench.json 2> /root/repo/gpurun_out/prof_bench.err
cd /root/repo; rm -f gpurun_out/prof/*agent_info* gpurun_out/prof/*kernel_trace*
cut -c1-220 gpurun_out/prof_bench.json; tail -2 gpurun_out/prof_bench.err
grep -E "av_row|qk_row|col_kernel|z_av|z_qk|linear_k" gpurun_out/prof/bench_kernel_stats.csv | cut -d, -f1-4
