#!/bin/bash
# x6: whole-tile ranges where the last round is nearly full (launch_x6 policy): checks, per-shape times, the step
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 300 python benchmarks/x6_bench.py --config vit_b16 --tiles 0 --iters 20 2>&1 | tail -30 ) > gpurun_out/s24_x6_bench.log
( timeout 200 python bench.py --steps 10 --cpu-baseline off > gpurun_out/s24_bench.json 2> gpurun_out/s24_bench.err )
cat gpurun_out/s24_x6_bench.log | cut -c1-420
cut -c1-300 gpurun_out/s24_bench.json; tail -3 gpurun_out/s24_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s24_bench.json").read().strip().splitlines()[-1])
for k in d["roofline"].get("kernels", [])[:8]:
    print(k["name"], k["launches"], round(k["avg_us"], 1), round(k["frac"], 3))
PY
