#!/usr/bin/env python
"""GPU measurement: achieved HBM rate of the streaming relprop kernels at the bench shapes (ViT-B/16, batch 64),
algorithmic bytes / HIP-event time, against the 8 TB/s peak."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import ops  # noqa: E402


def rate(fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    return us, nbytes / us / 1e6


def main():
    d = torch.device("cuda:0")
    B, H, N, C = 64, 12, 197, 768
    g, c = torch.randn(B, H, N, N, device=d), torch.randn(B, H, N, N, device=d)
    out = torch.empty(B, N, N, device=d)
    r, x0, x1 = (torch.randn(B, N, C, device=d) for _ in range(3))
    rows = [("gradcam_headmean", lambda: ops.gradcam_headmean(g, c, out=out), (2 * H + 1) * B * N * N * 4),
            ("add_relprop (sums + apply)", lambda: ops.add_relprop(r, x0, x1), 8 * B * N * C * 4),
            ("clone_relprop", lambda: ops.clone_relprop((r, x0), x1), 4 * B * N * C * 4)]
    for name, fn, nbytes in rows:
        us, tbs = rate(fn, nbytes)
        print(f"{name:28s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {tbs:5.2f} TB/s  {tbs / 8.0 * 100:5.1f} % of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
