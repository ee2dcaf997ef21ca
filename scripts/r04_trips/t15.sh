#!/bin/bash
# trip 15: in-step A/B of cheap knobs (same box, A B A B): Z-pass on 128 x 256 tiles (TE_X6_FLAGS=0x400), two steps in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t15_base.$rep.json 2> gpurun_out/t15_base.$rep.err
  TE_X6_FLAGS=0x400 timeout 300 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t15_z128.$rep.json 2> gpurun_out/t15_z128.$rep.err
  timeout 300 python bench.py --steps 10 --cpu-baseline off --inflight 2 > gpurun_out/t15_if2.$rep.json 2> gpurun_out/t15_if2.$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t15_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ks={k["name"]:k for k in d["roofline"]["kernels"]}
        print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2), {n: ks[n]["avg_us"] for n in ("linear_x6_cpass","linear_x6_zpass") if n in ks})
    except Exception as e: print(f, "failed", e)
PY
