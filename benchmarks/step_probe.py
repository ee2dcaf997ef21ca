#!/usr/bin/env python
"""Per-launch HIP-event times of ONE serial eager step of the bench workload, grouped by (kernel group, algorithmic work),
i.e. per layer shape -- the breakdown behind bench.py's roofline.kernels[] averages.

    python benchmarks/step_probe.py [--config vit_b16_224] [--steps 3] [--env TE_X6_FLAGS=...]
"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_b16_224")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    args = bench.parse_args(["--config", a.config, "--cpu-baseline", "off"])
    import transformer_explainability_amd as te
    from transformer_explainability_amd import ops
    te._lib.require_device()
    dev = torch.device("cuda:0")
    te.enable_tuned_gemms()
    ops.USE_FUSED_PRODUCERS = True
    wl = bench.Workload(args, 0, dev)
    timer = bench.KernelTimer()
    ops.KERNEL_TIMER = timer
    for _ in range(2):
        wl.eager_serial(*wl.inputs)
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.steps):
        wl.eager_serial(*wl.inputs)
    t1.record()
    torch.cuda.synchronize()
    ops.x6_raise_if_failed(dev)
    groups = collections.defaultdict(list)
    for name, flops, nbytes, s, e in timer.records:
        groups[(name, round(flops), round(nbytes))].append(s.elapsed_time(e) * 1e3)
    print(f"STEP {a.tag} serial eager step: {t0.elapsed_time(t1) / a.steps:.2f} ms")
    tot = collections.defaultdict(float)
    for (name, flops, nbytes), ts in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        ts.sort()
        per_step = sum(ts) / a.steps
        tot[name] += per_step
        if per_step > 150:
            print("GRP " + json.dumps(dict(tag=a.tag, name=name, gflop=round(flops / 1e9, 1), mb=round(nbytes / 1e6, 1),
                                           launches_per_step=len(ts) / a.steps, med_us=round(ts[len(ts) // 2], 1),
                                           per_step_us=round(per_step))))
    print("TOT " + json.dumps({k: round(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}))


if __name__ == "__main__":
    main()
