#!/bin/bash
# the rule reuses the |X| planes of the layer's forward product: tests, then the step A B A B (TE_X6_KEEP_ABS=0 disables the reuse)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_gpu_producers.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -x -k "producer or fused or config1 or dual or reuses or layer" 2>&1 | tail -6 ) > gpurun_out/s28_tests.log
for i in 1 2; do
  ( timeout 120 python bench.py --steps 20 --cpu-baseline off --no-roofline > gpurun_out/s28_reuse_$i.json 2> gpurun_out/s28_reuse_$i.err )
  ( TE_X6_KEEP_ABS=0 timeout 120 python bench.py --steps 20 --cpu-baseline off --no-roofline > gpurun_out/s28_noreuse_$i.json 2> gpurun_out/s28_noreuse_$i.err )
done
cat gpurun_out/s28_tests.log
for f in reuse_1 noreuse_1 reuse_2 noreuse_2; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/s28_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"], 1), "maps/s", round(d["ms_per_step"], 2), "ms/step; fp32-MFMA run", round(d["config"].get("fp32_mfma_ms_per_step", 0), 2))
except Exception as e:
    print("$f", "FAILED", e); print(open("gpurun_out/s28_$f.err").read()[-800:])
PY
done
