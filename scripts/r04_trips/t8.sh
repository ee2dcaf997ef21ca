#!/bin/bash
# trip 8: bench with the eager probe path warmed; A/B against the pins that reproduce the round-3 geometry (no split is not selectable: compare value with r3's 860-875)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
TE_BENCH_DUMP=1 timeout 400 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t8_bench_$i.json 2> gpurun_out/t8_bench_$i.err
grep -E "timed|comparison" gpurun_out/t8_bench_$i.err
done
python - <<'PY'
import json
d=json.loads(open("gpurun_out/t8_bench_2.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "maps/s", round(d["ms_per_step"],2), "ms", "fp32", round(d["config"].get("fp32_mfma_maps_per_s",0),1))
for k in d["roofline"]["kernels"][:12]:
    print("   ", k["name"], k["launches"], k["avg_us"], k["frac"])
PY
grep "probe linear" gpurun_out/t8_bench_2.err | cut -c18-150
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_producers.py -x -q -m gpu -k "soft_mask or bert_layout or bert_tiny or bert_base" > gpurun_out/t8_tests.log 2>&1
grep -v amdgpu gpurun_out/t8_tests.log | tail -6
