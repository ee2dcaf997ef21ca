#!/bin/bash
# SQ counters, L2 hit rate and HBM-side bytes of the x6 Linear.relprop kernels at the ViT-B batch-64 shapes (separate
# rocprofv3 --pmc passes with --kernel-trace only; traffic = 2 * FETCH_SIZE + WRITE_SIZE KB per the guide's gfx950 note).
#   gpurun --timeout 900 -- 'bash scripts/x6_pmc.sh'   -> gpurun_out/x6_pmc_summary.csv
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum"
P6="GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1)); rm -rf gpurun_out/x6pmc$i
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $P -f csv -d "$ROOT/gpurun_out/x6pmc$i" -o x6 -- \
      python "$ROOT/benchmarks/x6_study.py" --once > "$ROOT/gpurun_out/x6pmc$i.log" 2>&1 )
done
python - <<'PY'
import csv, glob, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/x6pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"x6_kernel<(\d), *(\d)", r["Kernel_Name"])
        if not m:
            m2 = re.search(r"split_kernel", r["Kernel_Name"])
            if not m2:
                continue
            key = "split"
        else:
            key = ("zpass" if m.group(2) == "0" else "cpass") + f"_wm{m.group(1)}_grid{r.get('Grid_Size','')}_lds{r.get('LDS_Block_Size','')}"
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = ["kernel,counter,mean_per_dispatch,dispatches"]
for k, cs in sorted(rows.items()):
    for c, v in sorted(cs.items()):
        out.append(f"{k},{c},{sum(v) / len(v):.6g},{len(v)}")
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f_kb, w_kb = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        out.append(f"{k},traffic_bytes(2*FETCH+WRITE KB),{(2 * f_kb + w_kb) * 1024:.6g},")
    if "TCC_HIT_sum" in cs:
        h, m = sum(cs["TCC_HIT_sum"]), sum(cs["TCC_MISS_sum"])
        out.append(f"{k},l2_hit_rate,{h / max(h + m, 1):.4g},")
    if "SQ_INSTS_VALU" in cs and "SQ_INSTS_MFMA" in cs:
        out.append(f"{k},valu_per_mfma,{sum(cs['SQ_INSTS_VALU']) / max(sum(cs['SQ_INSTS_MFMA']), 1):.4g},")
open("gpurun_out/x6_pmc_summary.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -3 gpurun_out/x6pmc1.log
