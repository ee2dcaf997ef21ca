#!/bin/bash
# study build: the step with the whole-tile policy of launch_x6 (TE_X6_SNAP unset) vs stream-K everywhere (TE_X6_SNAP=0), A B A B in one trip
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
  ( timeout 120 python bench.py --steps 20 --cpu-baseline off --no-roofline > gpurun_out/s25_policy_$i.json 2> gpurun_out/s25_policy_$i.err )
  ( TE_X6_SNAP=0 timeout 120 python bench.py --steps 20 --cpu-baseline off --no-roofline > gpurun_out/s25_streamk_$i.json 2> gpurun_out/s25_streamk_$i.err )
done
for f in policy_1 streamk_1 policy_2 streamk_2; do python - <<PY
import json
d = json.loads(open("gpurun_out/s25_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"], 1), "maps/s", round(d["ms_per_step"], 2), "ms/step; fp32-MFMA run", round(d["config"].get("fp32_mfma_ms_per_step", 0), 2))
PY
done
