#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -40 ) > gpurun_out/tests_full.log
( timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err )
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d /root/repo/gpurun_out/pmc_fetch -o p -- python /root/repo/benchmarks/linear_bench.py --skip-peak --reps 1 > /root/repo/gpurun_out/pmc_fetch.log 2>&1 )
( timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d /root/repo/gpurun_out/pmc_write -o p -- python /root/repo/benchmarks/linear_bench.py --skip-peak --reps 1 > /root/repo/gpurun_out/pmc_write.log 2>&1 )
cd /root/repo
rm -f gpurun_out/pmc_fetch/*agent_info* gpurun_out/pmc_write/*agent_info*
echo "=== tests ==="; cat gpurun_out/tests_full.log
echo "=== bench ==="; cat gpurun_out/bench_b64.json; tail -12 gpurun_out/bench_b64.err
