#!/usr/bin/env python
"""Debug aid: the QK rule against a torch fp64 evaluation, per output and row block."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import ops  # noqa: E402
d = torch.device("cuda:0")
for (B, H, N) in [(1, 2, 96), (1, 2, 96), (1, 2, 96), (2, 3, 197)]:
    D = 64
    torch.manual_seed(0)
    q, k = (torch.rand(B, H, N, D, device=d) + 0.05 for _ in range(2))
    z = q @ k.transpose(-1, -2)
    R = torch.rand(B, H, N, N, device=d) * 0.01
    cq, ck = ops.matmul_relprop_qk(R, q, k, out_scale=0.5, z=z)
    torch.cuda.synchronize()
    S = (R.double() / (z.double() + 1e-9))
    rq = (q.double() * (S @ k.double())) * 0.5
    rk = (k.double() * (S.transpose(-1, -2) @ q.double())) * 0.5
    for name, got, ref in (("cam_q", cq, rq), ("cam_k", ck, rk)):
        err = (got.double() - ref).abs()
        per_blk = [float(err[:, :, i:i + 32].max()) for i in range(0, N, 32)]
        print(f"B{B} H{H} N{N} {name}: max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e}) per row block " + " ".join(f"{x:.1e}" for x in per_blk), flush=True)
    if True:
        err = (cq.double() - rq).abs()
        bad = (err > 1e-6).nonzero()
        print("  bad cam_q elements:", bad.shape[0], "first:", bad[:6].tolist(), "rows:", sorted(set(bad[:, 2].tolist()))[:40], "d:", sorted(set(bad[:, 3].tolist()))[:70], flush=True)
        for _ in range(3):
            cq2, _ = ops.matmul_relprop_qk(R, q, k, out_scale=0.5, z=z)
            e2 = (cq2.double() - rq).abs()
            b2 = (e2 > 1e-6).nonzero()
            print("  rerun: bad", b2.shape[0], "bh/rows:", sorted(set((int(x[0]), int(x[1]), int(x[2]) // 32) for x in b2.tolist()))[:12], flush=True)
