#!/bin/bash
# study build: G-mode tile geometry A/B + in-kernel stamps
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for wm in 1 2; do
  ( TE_X6_G_WM=$wm timeout 100 python benchmarks/x6_gemm_bench.py 2>&1 | grep -E "GEMM|TOTAL" ) > gpurun_out/s20_wm$wm.log
  ( TE_X6_G_WM=$wm TE_X6_G_PROF=1 timeout 100 python benchmarks/x6_gemm_bench.py 2>&1 | grep -E "PROF" ) > gpurun_out/s20_prof_wm$wm.log
done
for wm in 1 2; do echo "== wm $wm"; python - <<PY
import json
for l in open("gpurun_out/s20_wm$wm.log"):
    if l.startswith("GEMM"):
        d = json.loads(l[5:]); print(d["layer"], d["direction"], d["x6_us"], d["x6_bf16_tf"])
    else: print(l.strip())
for l in open("gpurun_out/s20_prof_wm$wm.log"):
    print(l.strip()[:330])
PY
done
