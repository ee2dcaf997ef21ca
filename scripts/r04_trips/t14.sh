#!/bin/bash
# trip 14: same-box A/B of the 128 x 128 geometry policy in the step (TE_X6_SMALL_TILES=0 = round-3 geometries only); x6 tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -x -q -m gpu -k "x6 or linear" > gpurun_out/t14_tests.log 2>&1
grep -v amdgpu gpurun_out/t14_tests.log | tail -3
for rep in 1 2; do
for st in 1 0; do
  TE_X6_SMALL_TILES=$st timeout 300 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t14_vitb_small$st.$rep.json 2> gpurun_out/t14_vitb_small$st.$rep.err
  TE_X6_SMALL_TILES=$st timeout 300 python bench.py --config bert_base_512 --steps 6 --warmup 2 --cpu-baseline off > gpurun_out/t14_bert_small$st.$rep.json 2> gpurun_out/t14_bert_small$st.$rep.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t14_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ks={k["name"]:k for k in d["roofline"]["kernels"]}
        print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2), {n: ks[n]["avg_us"] for n in ("linear_x6_cpass","linear_x6_zpass","linear_forward_x6","linear_backward_x6") if n in ks})
    except Exception as e: print(f, "failed", e)
PY
