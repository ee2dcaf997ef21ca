#!/bin/bash
# trip 22: the concurrency-aware tile choice as bench default: ViT-B default run; BERT-512 and ViT-L/16-384 A/B (lib policy vs auto)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python bench.py --cpu-baseline off > gpurun_out/t22_vitb_default.json 2> gpurun_out/t22_vitb_default.err
for t in lib auto; do
  timeout 200 python bench.py --config bert_base_512 --steps 6 --warmup 2 --cpu-baseline off --x6-tile $t > gpurun_out/t22_bert_$t.json 2> gpurun_out/t22_bert_$t.err
done
for t in lib auto; do
  timeout 300 python bench.py --config vit_l16_384 --steps 3 --warmup 1 --cpu-baseline off --x6-tile $t > gpurun_out/t22_vitl_$t.json 2> gpurun_out/t22_vitl_$t.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/t22_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), d["unit"], round(d["ms_per_step"],2))
    except Exception as e: print(f, "failed", e)
PY
