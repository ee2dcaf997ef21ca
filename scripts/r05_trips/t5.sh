#!/bin/bash
# trip 5: effective shader clock under the attention kernels: GRBM_GUI_ACTIVE (cycles) / kernel duration (trace)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
rm -rf gpurun_out/clk
( cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d "$ROOT/gpurun_out/clk" -o attn -- \
    python "$ROOT/scripts/attn_bench.py" 64 12 197 64 producers > "$ROOT/gpurun_out/clk.log" 2>&1 )
python - <<'PY'
import csv, glob, collections, re
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Dispatch_Id"])] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob("gpurun_out/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name, d = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
        m = re.search(r"(av_kb_kernel<\d|qk_rule_kernel<\d|attn_fwd_kernel|av_rule_kernel<\d)", name)
        if m and d > 0:
            cnt[m.group(1)][r["Counter_Name"]].append((float(r["Counter_Value"]), d))
for k, cs in sorted(cnt.items()):
    for c, v in sorted(cs.items()):
        cyc = sum(x for x, _ in v) / len(v); ns = sum(d for _, d in v) / len(v)
        print(f"{k} {c}: {cyc:.4g} per dispatch, {ns/1e3:.1f} us under the profiler -> {cyc/ns:.3f} GHz (if the counter is per-chip cycles)")
PY
