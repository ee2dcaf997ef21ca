#!/bin/bash
# x6 rule test on every Linear shape of the three configurations + PMC traffic of the fp32-MFMA Linear kernels (comparison path)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
( timeout 300 python -m pytest tests/test_gpu_rules.py -m gpu -q -p no:cacheprovider -k "x6" 2>&1 | tail -4 ) > gpurun_out/s32_tests.log
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf gpurun_out/s32_p$i
  ( cd /tmp && TE_LINEAR_X6=0 timeout 120 rocprofv3 --kernel-trace --pmc $P -f csv -d "$ROOT/gpurun_out/s32_p$i" -o lin -- \
      python "$ROOT/benchmarks/x6_study.py" --once > "$ROOT/gpurun_out/s32_p$i.log" 2>&1 )
done
cat gpurun_out/s32_tests.log
python - <<'PY'
import csv, glob, collections, re, json
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/s32_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"linear_k([12])_kernel", r["Kernel_Name"])
        if m:
            rows["zpass_fwd" if m.group(1) == "1" else "cpass"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in rows.items():
    f_kb, w_kb = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    h, m_ = sum(cs["TCC_HIT_sum"]), sum(cs["TCC_MISS_sum"])
    out[f"vit_b16_b64.four_shapes.{k}"] = {"traffic_bytes": (2 * f_kb + w_kb) * 1024, "fetch_size_kb": f_kb, "write_size_kb": w_kb,
                                           "l2_hit_rate": h / max(h + m_, 1), "launches": len(cs["FETCH_SIZE"])}
json.dump(out, open("gpurun_out/s32_linear_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
