#!/bin/bash
# Generator(overlap_backward=True) for BERT: bitwise test, then the step off / on / off / on
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "beside_backward or bert_base_golden" 2>&1 | tail -4 ) > gpurun_out/s36_tests.log
for i in 1 2; do for ov in off on; do
  ( timeout 200 python bench.py --config bert_base_512 --steps 8 --warmup 2 --cpu-baseline off --no-roofline --overlap-backward $ov > gpurun_out/s36_${ov}_$i.json 2> gpurun_out/s36_${ov}_$i.err )
done; done
cat gpurun_out/s36_tests.log
for f in off_1 on_1 off_2 on_2; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/s36_$f.json").read().strip().splitlines()[-1])
    print("bert_base_512 overlap $f", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
except Exception as e:
    print("$f", "FAILED", e); print(open("gpurun_out/s36_$f.err").read()[-600:])
PY
done
