#!/bin/bash
# trip 24: whole-tile ranges always (TE_X6_FLAGS=0x10000) vs the policy, two steps in flight, same box A B A B; + the bitwise test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_rules.py -x -q -m gpu -k "schedules_are_bitwise" > gpurun_out/t24_tests.log 2>&1
grep -v amdgpu gpurun_out/t24_tests.log | tail -2
for rep in 1 2; do
  timeout 150 python bench.py --steps 12 --cpu-baseline off > gpurun_out/t24_base.$rep.json 2> gpurun_out/t24_base.$rep.err
  TE_X6_FLAGS=0x10000 timeout 150 python bench.py --steps 12 --cpu-baseline off > gpurun_out/t24_whole.$rep.json 2> gpurun_out/t24_whole.$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t24_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ks={k["name"]:k for k in d["roofline"]["kernels"]}
        print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2), {n: ks[n]["avg_us"] for n in ("linear_x6_cpass","linear_x6_zpass","linear_forward_x6","linear_backward_x6") if n in ks})
    except Exception as e: print(f, "failed", e)
PY
