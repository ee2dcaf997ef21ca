#!/usr/bin/env python
"""GPU tuning aid: shader-clock cycles per phase and tile of workgroup 0 of the attention rule kernels (TE_MARK hooks in
te_attn_rules.hip): 0 = tail of the previous tile .. loop top, 1 = barrier 1, 2 = S formation + LDS stores,
3 = barrier 2, 4 = issue of the next tile's loads, 5 = row-side product + epilogue, 6 = column-side product."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import _lib, ops  # noqa: E402

d = torch.device("cuda:0")
B, H, N, D = (int(a) for a in (sys.argv[1:5] + [64, 12, 197, 64][len(sys.argv) - 1:]))
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, device=d) for _ in range(3))
zqk = q @ k.transpose(-1, -2)
attn = torch.softmax(zqk * D ** -0.5, -1)
zav = attn @ v
R = torch.randn(B, H, N, D, device=d) * 0.01
Rnn = torch.randn(B, H, N, N, device=d) * 0.01
for _ in range(3):
    ops.matmul_relprop_av(R, attn, v, out_scale=0.5, z=zav)
    ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk)
torch.cuda.synchronize()
lib = _lib.load()
lib.te_attn_rules_profile.argtypes = [ctypes.c_void_p]
buf = torch.zeros(128, dtype=torch.int64, device=d)
lib.te_attn_rules_profile(buf.data_ptr())
ops.matmul_relprop_av(R, attn, v, out_scale=0.5, z=zav)
ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk)
torch.cuda.synchronize()
lib.te_attn_rules_profile(None)
ntiles = (N + 31) // 32
c = buf.cpu().view(2, 8, 8).double() / ntiles
for name, t in zip(("av_rule_kernel", "qk_rule_kernel"), c):
    print(name, "cycles per tile (rows = waves 0..7; columns = phases 0..6, total)")
    for w in range(8):
        print("  wave", w, " ".join(f"{x:8.0f}" for x in t[w][:7]), f"| {float(t[w][:7].sum()):8.0f}")
