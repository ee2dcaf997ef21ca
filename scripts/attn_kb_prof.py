#!/usr/bin/env python
"""Phase times of the AV kb kernel's workgroup 0 (library built with TE_BUILD_DEFINES=TE_STUDY, run with TE_ATTN_KB_PROF=1):
shader-clock cycles per wave, accumulated over the tiles of one launch.   python scripts/attn_kb_prof.py [B H N]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WHICH = sys.argv[4] if len(sys.argv) > 4 else "av"
os.environ["TE_ATTN_KB_PROF"] = "2" if WHICH == "qk" else "1"
from transformer_explainability_amd import _lib, ops  # noqa: E402

B, H, N = (int(a) for a in (sys.argv[1:4] + [64, 12, 197][len(sys.argv) - 1:]))
D = 64
d = torch.device("cuda:0")
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, device=d) for _ in range(3))
attn = torch.softmax((q @ k.transpose(-1, -2)) * D ** -0.5, -1)
zav = attn @ v
R = torch.randn(B, H, N, D, device=d) * 0.01
zqk = q @ k.transpose(-1, -2)
Rnn = torch.randn(B, H, N, N, device=d) * 0.01
for _ in range(3):
    if WHICH == "qk":
        ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk)
    else:
        ops.matmul_relprop_av(R, attn, v, out_scale=0.5, z=zav)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 64)()
fn = lib.te_attn_kb_prof_read
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
assert fn(buf) == 0
names = ["requests", "row product", "split+col product", "result+stores", "poll S", "to_acc+loop", "form S", "pro+epilogue"]
if WHICH == "qk":
    names = ["top(q planes,requests)", "sd+stage", "split+col", "rows+split+row", "wait readers+publish", "wait partials+fold", "to_acc x2", "pro+epilogue"]
ntiles = (N + 31) // 32
print(f"B={B} H={H} N={N}: {ntiles} tiles; cycles per wave of workgroup 0 (per tile in brackets)")
for w in range(8):
    row = [buf[w * 8 + i] for i in range(8)]
    tot = sum(row)
    print(f"wave {w}: total {tot:7d} | " + " | ".join(f"{n} {x} ({x // ntiles})" if i < 7 else f"{n} {x}" for i, (n, x) in enumerate(zip(names, row))))
