#!/bin/bash
# first GPU trip: parity tests, smoke, short bench, rocprof kernel stats
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_rules.py -m gpu -q -p no:cacheprovider 2>&1 | tail -150 ) > gpurun_out/test_rules.log
( timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider 2>&1 | tail -150 ) > gpurun_out/test_models.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -20 ) > gpurun_out/smoke.log
( timeout 600 python bench.py --steps 2 --warmup 1 --batch 16 --cpu-baseline off 2>&1 | tail -20 ) > gpurun_out/bench_b16.log
( timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -20 ) > gpurun_out/bench_b64.log
echo "=== rules ==="; tail -40 gpurun_out/test_rules.log
echo "=== models ==="; tail -40 gpurun_out/test_models.log
echo "=== smoke ==="; cat gpurun_out/smoke.log
echo "=== bench ==="; cat gpurun_out/bench_b16.log gpurun_out/bench_b64.log
