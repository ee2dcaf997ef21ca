#!/bin/bash
# SQ counters and HBM-side bytes of the attention rule / producer kernels at the bench shape (separate rocprofv3 --pmc
# passes, kernel-trace only; traffic = 2 * FETCH_SIZE + WRITE_SIZE KB per the guide's gfx950 correction, as in
# profiles/r01_linear_traffic_pmc.json).
#   gpurun --timeout 600 -- 'bash scripts/attn_pmc.sh'   -> gpurun_out/attn_pmc_summary.csv
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); rm -rf gpurun_out/pmc$i
  ( cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc $P -f csv -d "$ROOT/gpurun_out/pmc$i" -o attn -- \
      python "$ROOT/scripts/attn_bench.py" 64 12 197 64 producers > "$ROOT/gpurun_out/pmc$i.log" 2>&1 )
done
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "te_attn" not in k and "rule_kernel" not in k and "attn_fwd" not in k: continue
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/attn_pmc_summary.csv", "w") as out:
    out.write("kernel,counter,mean_per_dispatch,dispatches\n")
    for k, cs in sorted(rows.items()):
        for c, v in sorted(cs.items()):
            out.write(f"{k[-60:]},{c},{sum(v)/len(v):.6g},{len(v)}\n")
print(open("gpurun_out/attn_pmc_summary.csv").read())
PY
tail -3 gpurun_out/pmc1.log
