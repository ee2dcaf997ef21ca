cd "${GRAFT_REPO_ROOT:-.}"
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "producer or config2 or config3 or bert_base or cabi" 2>&1 | tail -8 ) > gpurun_out/tests_fwd6l.log
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline off > gpurun_out/bench_${cfg}_fwd6l.json 2> gpurun_out/bench_${cfg}_fwd6l.err )
  cut -c1-330 gpurun_out/bench_${cfg}_fwd6l.json; tail -2 gpurun_out/bench_${cfg}_fwd6l.err
done
cat gpurun_out/tests_fwd6l.log
