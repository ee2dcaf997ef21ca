#!/bin/bash
# full -m gpu suite with the round-3 defaults (x6 rules, x6 layer products, whole-tile policy) + the other two configurations
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 --maxfail=8 2>&1 | tail -45 ) > gpurun_out/tests_full.log
for cfg in vit_l16_384 bert_base_512; do
  for m in all off; do
    ( timeout 200 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline off --no-roofline --x6-gemm $m > gpurun_out/s27_${cfg}_$m.json 2> gpurun_out/s27_${cfg}_$m.err )
  done
done
tail -30 gpurun_out/tests_full.log
for f in vit_l16_384_all vit_l16_384_off bert_base_512_all bert_base_512_off; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/s27_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
except Exception as e:
    print("$f", "FAILED", e); print(open("gpurun_out/s27_$f.err").read()[-800:])
PY
done
