#!/bin/bash
# Round 6: L1 (TCP) -> L2 request counts and texture-addresser time of the x6 Z-pass (256 x 256 tiles) with its epilogue (opt0), without the
# epilogue's R / Y loads (opt16), without its S stores (opt32) and without the epilogue (opt4) -- measurement build, one process per
# variant, separate --pmc passes.   gpurun --timeout 900 -- 'bash scripts/x6_epi_pmc.sh'   -> gpurun_out/r06/x6_epi_pmc.csv
mkdir -p gpurun_out/r06; export TMPDIR=/tmp; ROOT=$PWD
export TE_RELPROP_LIB=$ROOT/transformer-explainability_amd/lib/libte_relprop_study.so TE_X6_SNAP=1
PA="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_READ_sum"
PB="TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
PC="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
for v in opt0 opt16 opt32 opt4; do
  i=0
  for P in "$PA" "$PB" "$PC"; do
    i=$((i+1)); rm -rf /tmp/epipmc_${v}_$i
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $P -f csv -d /tmp/epipmc_${v}_$i -o x6 -- \
        python "$ROOT/benchmarks/x6_variants.py" --variants $v --iters 1 > /tmp/epipmc_${v}_$i.log 2>&1 )
  done
done
python - <<'PY'
import csv, glob, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/epipmc_*/**/*counter_collection.csv", recursive=True):
    v = re.search(r"epipmc_(opt\d+)_", f).group(1)
    for r in csv.DictReader(open(f)):
        m = re.search(r"x6_kernel<(\d), *(\d),", r["Kernel_Name"])
        if m and m.group(2) == "0":
            rows[(v, "wm" + m.group(1), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = ["variant,geometry,grid,counter,sum_over_dispatches,dispatches"]
for k, cs in sorted(rows.items()):
    for c, vals in sorted(cs.items()):
        out.append(f"{k[0]},{k[1]},{k[2]},{c},{sum(vals):.6g},{len(vals)}")
open("gpurun_out/r06/x6_epi_pmc.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
