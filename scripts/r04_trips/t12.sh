#!/bin/bash
# trip 12: the lrp rule library (ViT_orig_LRP) as a bench line, A/B against ours in the same trip; --inflight 3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t12_bench_ours.json 2> gpurun_out/t12_bench_ours.err
timeout 400 python bench.py --steps 10 --cpu-baseline off --rules lrp > gpurun_out/t12_bench_lrp.json 2> gpurun_out/t12_bench_lrp.err
tail -3 gpurun_out/t12_bench_lrp.err
python - <<'PY'
import json
for n in ("ours","lrp"):
    d=json.loads(open(f"gpurun_out/t12_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), "maps/s", round(d["ms_per_step"],2), "ms", "fp32", round(d["config"].get("fp32_mfma_maps_per_s",0),1), d["roofline"]["kernel"][:60], round(d["roofline"]["frac"],3))
    for k in d["roofline"]["kernels"][:6]:
        print("   ", k["name"], k["launches"], k["avg_us"], k["frac"])
PY
TE_ALLOW_INFLIGHT=1 timeout 240 python bench.py --inflight 3 --steps 12 --cpu-baseline off > gpurun_out/t12_inflight3.json 2> gpurun_out/t12_inflight3.err
echo "inflight3 rc=$?"; grep -E "timed" gpurun_out/t12_inflight3.err; python -c "
import json; d=json.loads(open('gpurun_out/t12_inflight3.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2))"
