#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 --maxfail=10 2>&1 | tail -80 ) > gpurun_out/s7_tests_full.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/s7_smoke.log
( timeout 500 python bench.py > gpurun_out/s7_bench_b64.json 2> gpurun_out/s7_bench_b64.err )
echo "=== tests ==="; cat gpurun_out/s7_tests_full.log
echo "=== smoke ==="; cat gpurun_out/s7_smoke.log
echo "=== bench ==="; cut -c1-1500 gpurun_out/s7_bench_b64.json; tail -12 gpurun_out/s7_bench_b64.err
