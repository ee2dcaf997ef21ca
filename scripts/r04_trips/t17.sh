#!/bin/bash
# trip 17: sweep50k on one GPU with the batch-aware in-flight default (256 per step -> one step in flight), and at batch 32 (the 8-GPU share)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python bench.py --config sweep50k --steps 12 --cpu-baseline off > gpurun_out/t17_sweep.json 2> gpurun_out/t17_sweep.err
tail -2 gpurun_out/t17_sweep.err
timeout 400 python bench.py --config sweep50k --batch 32 --steps 40 --cpu-baseline off > gpurun_out/t17_sweep_b32.json 2> gpurun_out/t17_sweep_b32.err
tail -2 gpurun_out/t17_sweep_b32.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t17_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), d["unit"], round(d["ms_per_step"],2), "steps", d["steps"], "B", d["config"]["batch_per_gpu"], "inflight", d["config"]["steps_in_flight"])
    except Exception as e: print(f, "failed", e)
PY
