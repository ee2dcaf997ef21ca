#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/tests_full.log
echo "=== tests ==="; cat gpurun_out/tests_full.log
( timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -14 )
( timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline off > gpurun_out/bench_fwd.json 2> gpurun_out/bench_fwd.err )
echo "=== bench ==="; cut -c1-250 gpurun_out/bench_fwd.json; grep -o '"roofline".*' gpurun_out/bench_fwd.json | cut -c1-400; tail -3 gpurun_out/bench_fwd.err
