#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -30 ) > gpurun_out/s11_producers.log
( timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "config2 or config3 or bert_base_with_layer" 2>&1 | tail -30 ) > gpurun_out/s11_models.log
cat gpurun_out/s11_producers.log | cut -c1-250; cat gpurun_out/s11_models.log | cut -c1-250
for cfg in vit_l16_384 bert_base_512; do
  for pr in fused stock; do
    ( timeout 400 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline off --producers $pr > gpurun_out/s11_bench_${cfg}_$pr.json 2> gpurun_out/s11_bench_${cfg}_$pr.err )
    echo "== $cfg $pr"; cut -c1-200 gpurun_out/s11_bench_${cfg}_$pr.json; tail -2 gpurun_out/s11_bench_${cfg}_$pr.err
  done
done
