#!/usr/bin/env python
"""Phase stamps of the x6 Z-pass epilogue (measurement build, TE_X6_OPT = 8 | 64): shader-clock times of waves 0 and 4 (one SIMD) of
workgroup 163, per 32 x 32 block: wait for the block's R / Y (direct-to-LDS) loads, LDS read-back + request of block + 2, arithmetic,
store issue.   TE_RELPROP_LIB=.../libte_relprop_study.so python benchmarks/x6_epi_stamps.py [--layer qkv]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.x6_bench import CONFIGS, operands  # noqa: E402
from transformer_explainability_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="qkv")
a = ap.parse_args()
os.environ["TE_X6_OPT"] = str(8 | 64)
os.environ["TE_X6_SNAP"] = "1"
_lib.require_device()
lib = _lib.load()
dev = torch.device("cuda:0")
T, shapes = CONFIGS["vit_b16"]
al = lambda n: (n + 255) // 256 * 256      # noqa: E731
for (lname, in_f, out_f) in shapes:
    if lname != a.layer:
        continue
    X, W, b, R, Y = operands(T, in_f, out_f, 1, dev)
    planes = ops.x6_weight_planes(W, {})
    out = torch.empty_like(X)
    ws = torch.zeros(lib.te_linear_relprop_x6_workspace_bytes(T, in_f, out_f), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    off_flags = al(lib.te_linear_x6_planes_bytes(T, in_f)) + al(lib.te_linear_x6_planes_bytes(T, out_f)) + (512 * 256 * 128 * 4)
    for rep in range(3):
        for phase in (4, 8):
            _lib.check(lib.te_linear_relprop_x6_f32(R.data_ptr(), None, 0, 1, X.data_ptr(), W.data_ptr(), planes.data_ptr(),
                                                    None, Y.data_ptr(), b.data_ptr(), out.data_ptr(), T, in_f, out_f,
                                                    2 | phase, ops.x6_status(dev).data_ptr(), ws.data_ptr(), ws.numel(), st), "x6")
        torch.cuda.synchronize()
    raw = ws[off_flags + 8192 + 4608 * 8: off_flags + 8192 + 4608 * 8 + 2 * 64 * 8].view(torch.int64).view(2, 64).cpu()
    for wv in range(2):
        t = raw[wv].tolist()
        print(f"{lname}: wave {4 * wv} (last tile of workgroup 163): epilogue {t[41] - t[0]} cycles")
        names = ["wait R/Y", "read-back + request", "arithmetic", "stores + fallback check"]
        tot = [0, 0, 0, 0]
        for bi in range(8):
            s = t[1 + 5 * bi: 1 + 5 * bi + 5] + [t[1 + 5 * bi + 5] if bi < 7 else t[41]]
            d = [s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3]]
            for k in range(4):
                tot[k] += d[k]
            print(f"   block {bi}: " + "  ".join(f"{n} {v}" for n, v in zip(names, d)))
        print("   total:   " + "  ".join(f"{n} {v}" for n, v in zip(names, tot)))
