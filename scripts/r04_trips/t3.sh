#!/bin/bash
# trip 3: full -m gpu suite on the new kernels + step bench with the geometry policy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t3_bench.json 2> gpurun_out/t3_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/t3_bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "maps/s", round(d["ms_per_step"],2), "ms", "fp32", round(d["config"].get("fp32_mfma_maps_per_s",0),1))
for k in d["roofline"]["kernels"][:10]:
    print("   ", k["name"], k["launches"], k["avg_us"], k["frac"])
PY
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t3_pytest_gpu.log 2>&1
grep -v amdgpu gpurun_out/t3_pytest_gpu.log | tail -8
