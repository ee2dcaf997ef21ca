#!/bin/bash
# relprop beside the backward pass for the two other configurations, now that the backward products are x6 kernels too
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for cfg in vit_l16_384 bert_base_512; do
  for ov in off on; do
    ( timeout 200 python bench.py --config $cfg --steps 5 --warmup 2 --cpu-baseline off --no-roofline --overlap-backward $ov > gpurun_out/s35_${cfg}_$ov.json 2> gpurun_out/s35_${cfg}_$ov.err )
  done
done
for f in vit_l16_384_off vit_l16_384_on bert_base_512_off bert_base_512_on; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/s35_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step graph", d["config"]["hip_graph"])
except Exception as e:
    print("$f", "FAILED", e); print(open("gpurun_out/s35_$f.err").read()[-600:])
PY
done
