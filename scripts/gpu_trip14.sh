#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider -k "batch_equals or tiny" 2>&1 | tail -15 )
for g in on off; do
echo "=== graph $g ==="
( timeout 300 python bench.py --steps 8 --warmup 2 --cpu-baseline off --graph $g > gpurun_out/bench_g$g.json 2> gpurun_out/bench_g$g.err; tail -4 gpurun_out/bench_g$g.err; python - <<P
import json; d=json.load(open('gpurun_out/bench_g$g.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['launches_timed'], d['roofline']['zpass'])
P
)
done
