#!/bin/bash
# trip 21: under saturation (two steps in flight) what counts is CU-time per launch: every x6 launch on 256 x 256 tiles (TE_X6_FLAGS=2:
# narrow launches then occupy only 150 CUs, but each at full efficiency) vs the policy, same box A B A B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  timeout 200 python bench.py --steps 12 --cpu-baseline off > gpurun_out/t21_base.$rep.json 2> gpurun_out/t21_base.$rep.err
  TE_X6_FLAGS=2 timeout 200 python bench.py --steps 12 --cpu-baseline off > gpurun_out/t21_g256.$rep.json 2> gpurun_out/t21_g256.$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t21_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2))
    except Exception as e: print(f, "failed", e)
PY
