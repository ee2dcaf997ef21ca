#!/bin/bash
# trip 16: the new default (two replayed steps in flight) -- default bench x2, sweep50k, 2-rank rig, under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python bench.py > gpurun_out/t16_default.$rep.json 2> gpurun_out/t16_default.$rep.err
  grep -E "timed|comparison|captured" gpurun_out/t16_default.$rep.err
done
timeout 300 python bench.py --steps 20 --cpu-baseline off --inflight 1 > gpurun_out/t16_if1.json 2> gpurun_out/t16_if1.err
timeout 400 python bench.py --config sweep50k --steps 12 --cpu-baseline off > gpurun_out/t16_sweep.json 2> gpurun_out/t16_sweep.err
tail -2 gpurun_out/t16_sweep.err
TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 300 python bench.py --gpus 2 --batch 16 --steps 4 --warmup 2 --cpu-baseline off > gpurun_out/t16_rig2.json 2> gpurun_out/t16_rig2.err
tail -2 gpurun_out/t16_rig2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t16_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), d["unit"], round(d["ms_per_step"],2), "steps", d["steps"], "inflight", d["config"]["steps_in_flight"], "fp32", d["config"].get("fp32_mfma_maps_per_s"))
    except Exception as e: print(f, "failed", e)
PY
