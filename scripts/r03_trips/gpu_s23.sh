#!/bin/bash
# study build: stream-K cuts (0) vs ranges cut at tile boundaries only (1: whole tiles, rounds in k lock-step)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for o in 0 1; do
  ( TE_X6_SNAP=$o timeout 150 python benchmarks/x6_bench.py --config vit_b16 --tiles 0 --iters 10 --skip-checks 2>&1 | grep -E "BENCH|TOTAL" ) > gpurun_out/s23_rule_o$o.log
  ( TE_X6_SNAP=$o timeout 100 python benchmarks/x6_gemm_bench.py 2>&1 | grep -E "GEMM|TOTAL" ) > gpurun_out/s23_gemm_o$o.log
done
for o in 0 1; do echo "== snap $o"; python - <<PY
import json
for l in open("gpurun_out/s23_rule_o$o.log"):
    if l.startswith("BENCH"):
        d = json.loads(l[6:]); print("rule", d["layer"], "split", round(d["split_us"]), "z", round(d["z_us"]), "c", round(d["c_us"]), "rule", round(d["rule_us"]))
    else: print(l.strip())
for l in open("gpurun_out/s23_gemm_o$o.log"):
    if l.startswith("GEMM"):
        d = json.loads(l[5:]); print("gemm", d["layer"], d["direction"], d["x6_us"], d["x6_bf16_tf"])
    else: print(l.strip())
PY
done
