#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
val() { python -c "import json,sys;d=json.load(open('$1'));print('%.1f %s  %.2f ms/step'%(d['value'],d['unit'],d['ms_per_step']))" 2>/dev/null || echo "FAILED"; }
for t in auto 128 256; do
  ( timeout 150 python bench.py --cpu-baseline off --no-roofline --steps 12 --x6-tile $t > gpurun_out/s17_b_$t.json 2> gpurun_out/s17_b_$t.err )
  echo "vit_b16 x6-tile $t: $(val gpurun_out/s17_b_$t.json)"
done
for cfg in vit_l16_384 bert_base_512; do for pr in stock fused; do for ov in off on; do
  ( timeout 200 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline off --no-roofline --producers $pr --overlap-backward $ov > gpurun_out/s17_${cfg}_${pr}_$ov.json 2> gpurun_out/s17_${cfg}_${pr}_$ov.err )
  echo "$cfg producers $pr overlap $ov: $(val gpurun_out/s17_${cfg}_${pr}_$ov.json)"
done; done; done
