#!/bin/bash
# study build: HBM-side traffic and L2 hit rate of the x6 kernels under the two tile orders (separate --pmc passes)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
for o in 0 1; do
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); rm -rf gpurun_out/s22_o${o}_p$i
    ( cd /tmp && TE_X6_ORDER=$o timeout 120 rocprofv3 --kernel-trace --pmc $P -f csv -d "$ROOT/gpurun_out/s22_o${o}_p$i" -o x6 -- \
        python "$ROOT/benchmarks/x6_study.py" --once > "$ROOT/gpurun_out/s22_o${o}_p$i.log" 2>&1 )
  done
done
python - <<'PY'
import csv, glob, collections, re
for o in (0, 1):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/s22_o{o}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"x6_kernel<(\d), *(\d)", r["Kernel_Name"])
            if not m:
                continue
            key = ("zpass" if m.group(2) == "0" else "cpass") + f"_wm{m.group(1)}_grid{r.get('Grid_Size','')}"
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(rows.items()):
        line = f"order {o} {k}"
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f_kb, w_kb = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            line += f" traffic_bytes {(2 * f_kb + w_kb) * 1024:.4g} fetch_bytes {2 * f_kb * 1024:.4g} n {len(cs['FETCH_SIZE'])}"
        if "TCC_HIT_sum" in cs:
            h, m_ = sum(cs["TCC_HIT_sum"]), sum(cs["TCC_MISS_sum"])
            line += f" l2_hit {h / max(h + m_, 1):.3f}"
        print(line)
PY
