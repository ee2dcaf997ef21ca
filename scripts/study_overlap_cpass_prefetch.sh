#!/bin/bash
# C-pass prefetch-distance study (the TE_CPASS_PF variant it toggles was removed after this study: negative result)
# + relprop-beside-backward overlap, one trip.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
{
for pf in 1 2; do
  echo "== TE_CPASS_PF=$pf"
  TE_CPASS_PF=$pf python benchmarks/linear_bench.py --skip-peak --reps 5 2>&1 | grep -E "zfwd|cpass|block"
done
echo "== pytest (TE_CPASS_PF=2): linear rules + ViT-B batch/overlap equalities"
TE_CPASS_PF=2 python -m pytest tests/test_gpu_rules.py tests/test_gpu_models.py -q -m gpu -k "linear or batch_equals_singles or vit_tiny_golden" 2>&1 | tail -5
for ov in off on; do
  for pf in 1 2; do
    echo "== bench overlap=$ov TE_CPASS_PF=$pf"
    TE_CPASS_PF=$pf python bench.py --steps 8 --warmup 2 --cpu-baseline off --overlap-backward $ov 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('maps/s %.1f  ms/step %.2f  cpass %.1f TF frac %.3f  zfwd %.1f TF' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['zpass']['achieved']))"
  done
done
echo "== bench overlap=on eager (no graph)"
python bench.py --steps 8 --warmup 2 --cpu-baseline off --overlap-backward on --graph off 2>/dev/null | tail -1 | cut -c1-200
} 2>&1 | tee gpurun_out/trip_b.log
