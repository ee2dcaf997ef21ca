#!/bin/bash
# trip 1: new x6 tests (schedules bitwise, lost hand-over, two streams), stage / tile variants per shape, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rules.py -x -q -m gpu -k "x6" > gpurun_out/t1_tests.log 2>&1
tail -6 gpurun_out/t1_tests.log
timeout 400 python benchmarks/x6_variants.py --iters 10 --variants st2,st3,z128,c128,g128 > gpurun_out/t1_variants.log 2>&1
grep -v amdgpu.ids gpurun_out/t1_variants.log | tail -30
timeout 400 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t1_bench_st3.json 2> gpurun_out/t1_bench_st3.err
TE_X6_FLAGS=0x100 timeout 400 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t1_bench_st2.json 2> gpurun_out/t1_bench_st2.err
python - <<'PY'
import json
for n in ("st3","st2"):
    try:
        d=json.loads(open(f"gpurun_out/t1_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"],1), "maps/s", round(d["ms_per_step"],2), "ms", "fp32", round(d["config"].get("fp32_mfma_maps_per_s",0),1))
        for k in d["roofline"]["kernels"][:8]:
            print("   ", k["name"], k["launches"], k["avg_us"], k["frac"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 gpurun_out/t1_bench_st3.err
