#!/bin/bash
# LSU C-pass kernel: correctness + per-shape rate vs the tiled kernel + bench.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
show() { tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('maps/s %.1f  ms/step %.2f  cpass %.1f TF frac %.3f (%.1f us)  zfwd %.1f TF' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us'], r['zpass']['achieved']))
except Exception as e: print('FAILED', e)"; }
{
echo "== pytest linear + models (lsu)"
python -m pytest tests/test_gpu_rules.py tests/test_gpu_models.py -q -m gpu -x -k "linear or vit_tiny or bert_tiny or batch_equals_singles or golden_and_oracle" 2>&1 | tail -6
for k in tiled lsu; do
  echo "== microbench TE_CPASS_KERNEL=$k"
  TE_CPASS_KERNEL=$k python benchmarks/linear_bench.py --skip-peak --reps 5 2>&1 | grep -E "cpass|block"
done
for k in tiled lsu; do
  echo "== bench TE_CPASS_KERNEL=$k"
  TE_CPASS_KERNEL=$k python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>/dev/null | show
done
} 2>&1 | tee gpurun_out/trip_d.log
