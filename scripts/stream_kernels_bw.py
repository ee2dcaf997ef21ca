#!/usr/bin/env python
"""GPU measurement: achieved HBM rate of the streaming relprop kernels at the bench shapes (ViT-B/16, batch 64).

ALGORITHMIC bytes (Add 5 n, Clone 4 n, head-mean (2H+1) N^2 floats per sample) over HIP-event
time, against the 8 TB/s spec peak -- on COLD data: every repetition works on its own set of buffers and the sets
rotate through more than 1 GB, so nothing is served from the 256 MB Infinity Cache (the in-step condition: a rule's
operands were written by kernels that ran hundreds of MB of traffic earlier).

    python scripts/stream_kernels_bw.py                 # shipped kernels
    TE_HEADMEAN_VARIANT=0 python scripts/stream_kernels_bw.py --only headmean    # the grid-stride head-mean kernel
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import ops  # noqa: E402


def rate(make_call, sets, nbytes, reps=3):
    """make_call(i) runs the op on buffer set i.  One warm-up sweep, then `reps` timed sweeps over all sets."""
    n = len(sets)
    for i in range(n):
        make_call(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for i in range(n):
            make_call(i)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / (reps * n)
    return us, nbytes / us / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    d = torch.device("cuda:0")
    B, H, N, C = args.batch, 12, 197, 768
    rows = []

    def want(name):
        return not args.only or args.only in name

    if want("headmean"):
        # 2 x 119 MB per set; 6 sets = 1.4 GB
        sets = [(torch.randn(B, H, N, N, device=d), torch.randn(B, H, N, N, device=d), torch.empty(B, N, N, device=d))
                for _ in range(6)]
        rows.append(("gradcam_headmean", lambda i: ops.gradcam_headmean(sets[i][0], sets[i][1], out=sets[i][2]), sets,
                     (2 * H + 1) * B * N * N * 4))
    if want("add") or want("clone"):
        # 3 x 39 MB inputs per set; 12 sets = 1.4 GB
        esets = [tuple(torch.randn(B, N, C, device=d) for _ in range(3)) for _ in range(12)]
        fac = (torch.rand(B, 2, device=d) + 0.5)
        if want("add"):
            rows.append(("add_relprop two-pass (sums + apply)", lambda i: ops.add_relprop(*esets[i]), esets,
                         5 * B * N * C * 4))
            rows.append(("add_relprop deferred (one pass + factors)", lambda i: ops.add_relprop(*esets[i], deferred=True),
                         esets, 5 * B * N * C * 4))
        if want("clone"):
            rows.append(("clone_relprop (2 aliases)", lambda i: ops.clone_relprop((esets[i][0], esets[i][1]), esets[i][2]),
                         esets, 4 * B * N * C * 4))
            rows.append(("clone_relprop (deferred factor on R0)",
                         lambda i: ops.clone_relprop((ops.Deferred(esets[i][0], fac[:, 0]), esets[i][1]), esets[i][2]),
                         esets, 4 * B * N * C * 4))
    if want("rollout"):
        L = 12
        rsets = [torch.rand(L, B, N, N, device=d) * 0.01 for _ in range(8)]     # 119 MB each
        rows.append(("rollout row-0 chain (11 layers)", lambda i: ops.rollout(rsets[i], start_layer=1, row0_only=True),
                     rsets, (L - 1) * B * N * N * 4))
    for name, fn, sets, nbytes in rows:
        us, tbs = rate(fn, sets, nbytes)
        print(f"{name:44s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB algorithmic  {tbs:5.2f} TB/s  "
              f"{tbs / 8.0 * 100:5.1f} % of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
