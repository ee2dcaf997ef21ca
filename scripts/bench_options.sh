#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
show() { tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('maps/s %.1f  ms/step %.2f  cpass %.1f TF frac %.3f  zfwd %.1f TF' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['zpass']['achieved']))
except Exception as e: print('FAILED', e)"; }
{
echo "== pytest: pruning / overlap / tuned gemms / segmentation / sweep"
python -m pytest tests -q -m gpu -k "batch_equals_singles or pruned or tuned or segmentation or sweep or full_batch" 2>&1 | tail -5
echo "== bench default"; python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>/dev/null | show
echo "== bench --prune on"; python bench.py --steps 8 --warmup 2 --cpu-baseline off --prune on 2>/dev/null | show
echo "== bench --overlap-backward on"; python bench.py --steps 8 --warmup 2 --cpu-baseline off --overlap-backward on 2>/dev/null | show
echo "== bench --prune on --overlap-backward on"; python bench.py --steps 8 --warmup 2 --cpu-baseline off --prune on --overlap-backward on 2>/dev/null | show
} 2>&1 | tee gpurun_out/trip_f.log
