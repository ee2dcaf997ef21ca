#!/bin/bash
# trip 19: with two steps in flight (the GPU saturated), same box A B A B: the 128 x 128 policy on / off; relprop beside backward on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python bench.py --steps 12 --cpu-baseline off > gpurun_out/t19_base.$rep.json 2> gpurun_out/t19_base.$rep.err
  TE_X6_SMALL_TILES=0 timeout 300 python bench.py --steps 12 --cpu-baseline off > gpurun_out/t19_nosmall.$rep.json 2> gpurun_out/t19_nosmall.$rep.err
  timeout 300 python bench.py --steps 12 --cpu-baseline off --overlap-backward off > gpurun_out/t19_noov.$rep.json 2> gpurun_out/t19_noov.$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t19_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2), "inflight", d["config"]["steps_in_flight"])
    except Exception as e: print(f, "failed", e)
PY
