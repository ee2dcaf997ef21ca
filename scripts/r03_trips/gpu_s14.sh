#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 ) > gpurun_out/s14_producers.log
cat gpurun_out/s14_producers.log | cut -c1-300
( timeout 500 python bench.py --cpu-baseline off > gpurun_out/s14_bench_b64.json 2> gpurun_out/s14_bench_b64.err )
cut -c1-260 gpurun_out/s14_bench_b64.json; tail -3 gpurun_out/s14_bench_b64.err
( TE_X6_GEMM=0 timeout 500 python bench.py --cpu-baseline off > gpurun_out/s14_bench_b64_stockgemm.json 2> gpurun_out/s14_bench_b64_stockgemm.err )
cut -c1-260 gpurun_out/s14_bench_b64_stockgemm.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/s14_bench_b64.json'))
for k in d['roofline']['kernels']: print('   ',k['name'],k['launches'],k['avg_us'],'%.2f ms'%(k['launches']*k['avg_us']/1e3),k['frac'])
PY
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 400 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline off > gpurun_out/s14_bench_${cfg}.json 2> gpurun_out/s14_bench_${cfg}.err )
  echo "== $cfg"; cut -c1-200 gpurun_out/s14_bench_${cfg}.json; tail -2 gpurun_out/s14_bench_${cfg}.err
done
