#!/bin/bash
# SQ counters and HBM-side bytes of the attention rule / producer kernels at the bench shape (separate rocprofv3 --pmc
# passes, kernel-trace only; traffic = 2 * FETCH_SIZE + WRITE_SIZE KB per the guide's gfx950 correction, as in
# profiles/r01_linear_traffic_pmc.json).
#   gpurun --timeout 600 -- 'bash scripts/attn_pmc.sh'   -> gpurun_out/attn_pmc_summary.csv
#   ATTN_PMC_SHAPE="32 16 577" ATTN_PMC_TAG=_n577 bash scripts/attn_pmc.sh   -> the long-sequence kernels (te_attn_fwd6l / te_attn_bwd6l)
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
SHAPE=${ATTN_PMC_SHAPE:-64 12 197}; export ATTN_PMC_SHAPE="$SHAPE" ATTN_PMC_TAG
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
PASSES=("$P1" "$P2" "$P3" "$P4"); [ -n "$ATTN_PMC_SKIP_P2" ] && PASSES=("$P1" "$P3" "$P4")
for P in "${PASSES[@]}"; do
  i=$((i+1)); rm -rf gpurun_out/pmc$i
  ( cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc $P -f csv -d "$ROOT/gpurun_out/pmc$i" -o attn -- \
      python "$ROOT/scripts/attn_bench.py" $SHAPE 64 producers > "$ROOT/gpurun_out/pmc$i.log" 2>&1 )
done
python - <<'PY'
import csv, glob, collections, os, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(av6?_kb_kernel<\d|qk6?_kb_kernel<\d|fwd6l_kernel|bwd6l_rows_kernel|bwd6l_cols_kernel|bwd_rows_kernel|bwd_cols_kernel|fwd_rows_kernel|av_rule_kernel<\d|qk_rule_kernel<\d|qk_rc_kernel<\d|fwd6_kernel|attn_fwd_kernel|qk_finish_kernel)", r["Kernel_Name"])
        if m:
            k = m.group(1).replace("av6_", "av_").replace("qk6_", "qk_")
            rows[k + (">" if "<" in k else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
B, H, N = (int(x) for x in os.environ["ATTN_PMC_SHAPE"].split())      # scripts/attn_bench.py's shape; <0> = relprop rule, <1> = the backward producer on the same kernel
nn, nc = 2 * H * N * N, N * H * 64
alg = {"av_kb_kernel<0>": (nn + 5 * nc) * 4 * B, "av_kb_kernel<1>": (nn + 4 * nc) * 4 * B, "qk_kb_kernel<0>": (nn + 4 * nc) * 4 * B, "qk_kb_kernel<1>": (nn + 4 * nc) * 4 * B, "fwd_rows_kernel": (nn + 4 * nc) * 4 * B, "av_rule_kernel<0>": (nn + 5 * nc) * 4 * B, "qk_rule_kernel<0>": (nn + 4 * nc) * 4 * B,
       "attn_fwd_kernel": (nn + 4 * nc) * 4 * B, "fwd6_kernel": (nn + 4 * nc) * 4 * B, "qk_rc_kernel<0>": (nn + 4 * nc) * 4 * B, "qk_rc_kernel<1>": (nn + 4 * nc) * 4 * B, "av_rule_kernel<1>": (nn + 4 * nc) * 4 * B,
       "qk_rule_kernel<1>": (nn + 4 * nc) * 4 * B,
       # beyond 224 tokens: forward = two N x N out, q k v in, out out; backward rows = attn in, d_attn out, d_out out k v in, d_q out;
       # backward cols = attn and d_attn in, d_out q in, d_v d_k out
       "fwd6l_kernel": (nn + 4 * nc) * 4 * B, "bwd6l_rows_kernel": (nn + 5 * nc) * 4 * B, "bwd6l_cols_kernel": (nn + 4 * nc) * 4 * B}
out = ["kernel,counter,mean_per_dispatch,dispatches"]
for k, cs in sorted(rows.items()):
    for c, v in sorted(cs.items()):
        out.append(f"{k},{c},{sum(v) / len(v):.6g},{len(v)}")
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f_kb, w_kb = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        out.append(f"{k},traffic_bytes(2*FETCH+WRITE KB),{(2 * f_kb + w_kb) * 1024:.6g},")
        out.append(f"{k},algorithmic_bytes,{alg.get(k, 0):.6g},")
    if "SQ_INSTS_VALU" in cs and "SQ_INSTS_MFMA" in cs:
        out.append(f"{k},valu_per_mfma,{sum(cs['SQ_INSTS_VALU']) / sum(cs['SQ_INSTS_MFMA']):.4g},")
open("gpurun_out/attn_pmc_summary%s.csv" % os.environ.get("ATTN_PMC_TAG", ""), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -3 gpurun_out/pmc1.log
