#!/usr/bin/env python
"""Who owns the wall time of the overlapped step?  Reads a rocprofv3 --kernel-trace CSV of `bench.py` and, over the last
`--frac` of the trace (the timed graph replays), reports: wall span, time with no kernel running, time with an x6_kernel
running, time when ONLY kernels of a class run ("exclusive"), and each class's summed duration -- classes by kernel name.

    rocprofv3 --kernel-trace -f csv -d out -o bench -- python bench.py --steps 6 --cpu-baseline off
    python scripts/timeline_stats.py out/bench_kernel_trace.csv
"""
import argparse
import csv
import re
import sys

CLASSES = [
    ("x6_C", r"x6_kernel<\d, 1,"), ("x6_Z", r"x6_kernel<\d, 0,"), ("x6_G", r"x6_kernel<\d, 2,"), ("x6_other", r"x6_kernel"),
    ("attn_fwd", r"attn_fwd"), ("qk_rule", r"qk_r|qk_rule"), ("av", r"av6_kb|av_rule"), ("attn_bwd", r"attn_bwd|softmax_bwd"),
    ("split", r"split_kernel|gelu_split"), ("gelu", r"gelu"), ("ln", r"ln_(fwd|bwd)"), ("add/clone", r"add_|clone"),
    ("headmean/rollout", r"headmean|rollout"), ("zero", r"zero_words"), ("torch", r"at::native|Cijk|igemm"),
]


def cls(name):
    for c, pat in CLASSES:
        if re.search(pat, name):
            return c
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--ms", type=float, default=300.0, help="analyse the last MS milliseconds of the last dense burst of x6 kernels")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"])))
    rows.sort()
    # the timed region = the last burst of kernels around x6 launches with no gap above 20 ms; keep its last --ms milliseconds
    last_x6 = max(i for i, r in enumerate(rows) if r[2].startswith("x6"))
    i = last_x6
    while i > 0 and rows[i][0] - rows[i - 1][1] < 20e6:
        i -= 1
    j = last_x6
    while j + 1 < len(rows) and rows[j + 1][0] - rows[j][1] < 20e6:
        j += 1
    rows = rows[i:j + 1]
    hi = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= hi - a.ms * 1e6 - 30e6 and r[1] <= hi - 30e6]      # (and drop the burst's last 30 ms: the drain)
    ev = []
    for s, e, c in rows:
        ev.append((s, 1, c))
        ev.append((e, -1, c))
    ev.sort()
    live = {}
    last = ev[0][0]
    span = ev[-1][0] - ev[0][0]
    idle = 0
    excl, anyc, conc = {}, {}, 0
    for t, d, c in ev:
        dt = t - last
        if dt > 0:
            act = [k for k, v in live.items() if v > 0]
            if not act:
                idle += dt
            else:
                if len(act) == 1:
                    excl[act[0]] = excl.get(act[0], 0) + dt
                for k in act:
                    anyc[k] = anyc.get(k, 0) + dt
                if sum(live.values()) > 1:
                    conc += dt
        live[c] = live.get(c, 0) + d
        last = t
    dur = {}
    for s, e, c in rows:
        dur[c] = dur.get(c, 0) + (e - s)
    x6 = [k for k in anyc if k.startswith("x6")]
    print(f"span {span/1e6:.2f} ms, {len(rows)} kernels; idle {idle/1e6:.2f} ms ({100*idle/span:.1f} %), >= 2 kernels in flight {conc/1e6:.2f} ms ({100*conc/span:.1f} %)")
    print(f"{'class':18s} {'sum of durations':>18s} {'wall with it live':>18s} {'wall ONLY it live':>18s}   (ms, % of span)")
    for k in sorted(dur, key=lambda k: -dur[k]):
        print(f"{k:18s} {dur[k]/1e6:10.2f} {100*dur[k]/span:6.1f}% {anyc.get(k,0)/1e6:10.2f} {100*anyc.get(k,0)/span:6.1f}% {excl.get(k,0)/1e6:10.2f} {100*excl.get(k,0)/span:6.1f}%")


if __name__ == "__main__":
    sys.exit(main())
