#!/bin/bash
# per-direction x6 GEMM policy (forward / input gradient of the Linear layers): producer tests, then the step A B A B
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 ) > gpurun_out/s26_tests.log
for i in 1 2; do
  for m in auto off; do
    ( timeout 120 python bench.py --steps 20 --cpu-baseline off --no-roofline --x6-gemm $m > gpurun_out/s26_${m}_$i.json 2> gpurun_out/s26_${m}_$i.err )
  done
done
( timeout 120 python bench.py --steps 20 --cpu-baseline off --no-roofline --x6-gemm all > gpurun_out/s26_all_1.json 2> gpurun_out/s26_all_1.err )
cat gpurun_out/s26_tests.log
for f in auto_1 off_1 auto_2 off_2 all_1; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/s26_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"], 1), "maps/s", round(d["ms_per_step"], 2), "ms/step; fp32-MFMA run", round(d["config"].get("fp32_mfma_ms_per_step", 0), 2))
except Exception as e:
    print("$f", "FAILED", e); print(open("gpurun_out/s26_$f.err").read()[-600:])
PY
done
