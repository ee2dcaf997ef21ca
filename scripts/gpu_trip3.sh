#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
( timeout 200 python benchmarks/linear_bench.py 2>&1 | tail -20 ) > gpurun_out/linbench_auto.log
( TE_LINEAR_BN=128 timeout 200 python benchmarks/linear_bench.py --skip-peak 2>&1 | tail -12 ) > gpurun_out/linbench_128.log
( TE_LINEAR_BN=64 timeout 200 python benchmarks/linear_bench.py --skip-peak 2>&1 | tail -12 ) > gpurun_out/linbench_64.log
( timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/tests.log
for s in 1 2 3; do
( timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off --streams $s > gpurun_out/bench_s$s.json 2> gpurun_out/bench_s$s.err )
done
echo "=== linbench auto ==="; cat gpurun_out/linbench_auto.log
echo "=== linbench 128 ==="; cat gpurun_out/linbench_128.log
echo "=== linbench 64 ==="; cat gpurun_out/linbench_64.log
echo "=== tests ==="; tail -25 gpurun_out/tests.log
for s in 1 2 3; do echo "=== bench streams $s ==="; cut -c1-400 gpurun_out/bench_s$s.json; grep -o '"roofline".*' gpurun_out/bench_s$s.json | cut -c1-300; tail -3 gpurun_out/bench_s$s.err; done
