#!/bin/bash
# trip 23: the final headline line (final defaults: two steps in flight, large tiles under concurrency), 20 steps, with the CPU baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 > gpurun_out/t23_bench_b64.json 2> gpurun_out/t23_bench_b64.err
echo "rc=$?"; grep -E "timed|comparison" gpurun_out/t23_bench_b64.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/t23_bench_b64.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],2), d["config"]["steps_in_flight"], d["config"].get("fp32_mfma_maps_per_s"), d["roofline"]["frac"], d["roofline"]["traffic"])
for k in d["roofline"]["kernels"][:9]: print("  ", k["name"], k["launches"], k["avg_us"], k["frac"])
PY
