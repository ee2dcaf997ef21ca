#!/bin/bash
# split kernel with row-contiguous loads through LDS: plane tests (vs a torch restatement), x6 rule tests, split times per shape
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_producers.py tests/test_gpu_rules.py -m gpu -q -p no:cacheprovider -k "x6 or split or reuses or linear" 2>&1 | tail -5 ) > gpurun_out/s33_tests.log
( timeout 100 python benchmarks/x6_gemm_bench.py 2>&1 | grep -E "GEMM|TOTAL" ) > gpurun_out/s33_gemm.log
cat gpurun_out/s33_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/s33_gemm.log"):
    if l.startswith("GEMM"):
        d = json.loads(l[5:]); print(d["layer"], d["direction"], "K", d["K"], "split", d["split_us"], "x6", d["x6_us"])
    else: print(l.strip())
PY
