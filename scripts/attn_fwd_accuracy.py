#!/usr/bin/env python
"""Accuracy of the attention forward producer against fp64 (rms / max error of z_qk, attn, out), for the kernel the library selects
(measurement build: TE_ATTN_FWD=old = the round-2 kernel) and for stock PyTorch fp32 on the same device."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_explainability_amd import ops  # noqa: E402

d = torch.device("cuda:0")
B, H, N, D = 8, 12, 197, 64
torch.manual_seed(3)
for name, mult in (("randn", 1.0), ("randn x 3", 3.0)):
    qkv = torch.randn(B, N, 3 * H * D, device=d) * mult
    scale = D ** -0.5
    out, attn, zqk = ops.attention_forward(qkv, H, scale)
    q64 = qkv.double()
    q, k, v = q64.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    z64 = q @ k.transpose(-1, -2)
    a64 = torch.softmax(z64 * scale, -1)
    o64 = (a64 @ v).permute(0, 2, 1, 3).reshape(B, N, H * D)
    q32, k32, v32 = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    z32 = q32 @ k32.transpose(-1, -2)
    a32 = torch.softmax(z32 * scale, -1)
    o32 = (a32 @ v32).permute(0, 2, 1, 3).reshape(B, N, H * D)

    def err(x, r):
        e = (x.double() - r)
        return f"rms {float(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()):.3e} max {float(e.abs().max() / r.abs().max()):.3e} relmax {float((e.abs() / r.abs().clamp_min(1e-30)).max()):.3e}"
    print(name, os.environ.get("TE_ATTN_FWD", "new"))
    print("  ours  z", err(zqk, z64), "| attn", err(attn, a64), "| out", err(out, o64))
    print("  torch z", err(z32, z64), "| attn", err(a32, a64), "| out", err(o32, o64))
