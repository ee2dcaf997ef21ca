#!/bin/bash
# trip 13: same-box A/B of the K split in the step: ViT-B/16 and BERT-512 (the split is a property of the layer shape, the
# same for both; T differs: 12 608 vs 16 384 rows)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for ks in 1 0; do
  TE_X6_KSPLIT=$ks timeout 300 python bench.py --config bert_base_512 --steps 6 --warmup 2 --cpu-baseline off > gpurun_out/t13_bert_ks$ks.$rep.json 2> gpurun_out/t13_bert_ks$ks.$rep.err
  TE_X6_KSPLIT=$ks timeout 300 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t13_vitb_ks$ks.$rep.json 2> gpurun_out/t13_vitb_ks$ks.$rep.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t13_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ks={k["name"]:k for k in d["roofline"]["kernels"]}
        print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2), {n: ks[n]["avg_us"] for n in ("linear_x6_cpass","linear_x6_zpass","linear_forward_x6","linear_backward_x6") if n in ks})
    except Exception as e: print(f, "failed", e)
PY
