#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 1 2 3; do
( timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline off --inflight $n > gpurun_out/bench_if$n.json 2> gpurun_out/bench_if$n.err )
echo "=== inflight $n ==="; python - <<P
import json; d=json.load(open('gpurun_out/bench_if$n.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['zpass'])
P
tail -2 gpurun_out/bench_if$n.err
done
