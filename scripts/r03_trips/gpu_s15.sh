#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for inf in 1 2 3; do for ov in off on; do
  ( timeout 300 python bench.py --cpu-baseline off --steps 12 --warmup 2 --inflight $inf --overlap-backward $ov > gpurun_out/s15_inf${inf}_$ov.json 2> gpurun_out/s15_inf${inf}_$ov.err )
  echo "inflight $inf overlap $ov: $(python -c "import json;d=json.load(open('gpurun_out/s15_inf${inf}_$ov.json'));print('%.1f maps/s %.2f ms'%(d['value'],d['ms_per_step']))" 2>&1 | tail -1)"
done; done
( timeout 300 python bench.py --cpu-baseline off --steps 12 --warmup 2 --inflight 2 --no-roofline > gpurun_out/s15_inf2_noroof.json 2> gpurun_out/s15_inf2_noroof.err )
python -c "import json;d=json.load(open('gpurun_out/s15_inf2_noroof.json'));print('inflight 2 no probe step: %.1f maps/s'%d['value'])"
( timeout 300 python bench.py --cpu-baseline off --steps 12 --warmup 2 --no-roofline > gpurun_out/s15_inf1_noroof.json 2> gpurun_out/s15_inf1_noroof.err )
python -c "import json;d=json.load(open('gpurun_out/s15_inf1_noroof.json'));print('inflight 1 no probe step: %.1f maps/s'%d['value'])"
tail -3 gpurun_out/s15_inf3_on.err
