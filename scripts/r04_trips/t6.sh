#!/bin/bash
# trip 6: K split as its own instantiation: tests, variants, step bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -x -q -m gpu -k "x6 or linear or bert_layout" > gpurun_out/t6_tests.log 2>&1
grep -v amdgpu gpurun_out/t6_tests.log | tail -5
timeout 300 python benchmarks/x6_variants.py --iters 10 --variants base,g128,g64,g256 > gpurun_out/t6_variants_vitb.log 2>&1
grep -v amdgpu.ids gpurun_out/t6_variants_vitb.log | cut -c1-215 | tail -22
timeout 300 python benchmarks/x6_variants.py --iters 6 --variants base,g128,g64,g256 --config bert_base > gpurun_out/t6_variants_bert.log 2>&1
grep -v amdgpu.ids gpurun_out/t6_variants_bert.log | cut -c1-215 | tail -16
timeout 400 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t6_bench.json 2> gpurun_out/t6_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/t6_bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "maps/s", round(d["ms_per_step"],2), "ms", "fp32", round(d["config"].get("fp32_mfma_maps_per_s",0),1))
for k in d["roofline"]["kernels"][:8]:
    print("   ", k["name"], k["launches"], k["avg_us"], k["frac"])
PY
