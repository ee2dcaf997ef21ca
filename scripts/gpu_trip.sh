#!/bin/bash
# GPU trip: diagnostics, parity tests, bench with roofline + cpu_baseline, rocprofv3 kernel stats.
# Every leg has its own timeout; logs land in gpurun_out/.
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
nproc > gpurun_out/host.log; cat /sys/fs/cgroup/cpu.max >> gpurun_out/host.log 2>&1; free -g >> gpurun_out/host.log
( timeout 300 python scripts/debug_batch.py 4 7 2>&1 | tail -80 ) > gpurun_out/debug_batch.log
( timeout 600 python -m pytest tests/test_gpu_rules.py -m gpu -q -p no:cacheprovider 2>&1 | tail -120 ) > gpurun_out/test_rules.log
( timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider 2>&1 | tail -150 ) > gpurun_out/test_models.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -20 ) > gpurun_out/smoke.log
( timeout 420 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err )
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d /root/repo/gpurun_out/prof -o bench -- \
    python /root/repo/bench.py --steps 2 --warmup 1 --cpu-baseline off > /root/repo/gpurun_out/prof_bench.json 2> /root/repo/gpurun_out/prof_bench.err )
find gpurun_out/prof -name '*agent_info*' -delete 2>/dev/null
find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
echo "=== debug ==="; cat gpurun_out/debug_batch.log
echo "=== rules ==="; tail -15 gpurun_out/test_rules.log
echo "=== models ==="; tail -30 gpurun_out/test_models.log
echo "=== smoke ==="; cat gpurun_out/smoke.log
echo "=== bench ==="; cat gpurun_out/bench_b64.json; tail -20 gpurun_out/bench_b64.err
echo "=== prof ==="; cat gpurun_out/prof_bench.json; tail -5 gpurun_out/prof_bench.err; ls -R gpurun_out/prof | head
