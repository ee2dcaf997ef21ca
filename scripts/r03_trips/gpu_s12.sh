#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -12 ) > gpurun_out/s12_producers.log
cat gpurun_out/s12_producers.log | cut -c1-250
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 400 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline off --producers fused > gpurun_out/s12_bench_${cfg}_fused.json 2> gpurun_out/s12_bench_${cfg}_fused.err )
  echo "== $cfg fused"; cut -c1-200 gpurun_out/s12_bench_${cfg}_fused.json; tail -2 gpurun_out/s12_bench_${cfg}_fused.err
  python - <<PY
import json
d=json.load(open('gpurun_out/s12_bench_${cfg}_fused.json'))
for k in d['roofline']['kernels']:
    if k['name'].startswith('attention'): print('   ',k['name'],k['launches'],k['avg_us'],k['frac'])
PY
done
