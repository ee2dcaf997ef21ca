#!/usr/bin/env python
"""Measurement build only (TE_BUILD_DEFINES=TE_STUDY): phase stamps of one second-round workgroup of the row / key-block-owner
QK kernel (csrc/te_attn_rc.hip), per wave, in shader-clock cycles; plus HIP-event times of the rule and of the producers."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import _lib, ops  # noqa: E402


def main():
    d = torch.device("cuda:0")
    B, H, N, D = 64, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 197, 64
    torch.manual_seed(0)
    q, k = (torch.randn(B, H, N, D, device=d) for _ in range(2))
    zqk = q @ k.transpose(-1, -2)
    Rnn = torch.randn(B, H, N, N, device=d) * 0.01
    for _ in range(3):
        ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk)
    e.record()
    torch.cuda.synchronize()
    print(f"QK rule N={N}: {s.elapsed_time(e) * 100:.1f} us per launch")
    lib = _lib.load()
    if not hasattr(lib, "te_attn_rc_profile"):
        print("(not a study build: no stamps)")
        return
    buf = (ctypes.c_longlong * 128)()
    lib.te_attn_rc_profile.argtypes = [ctypes.c_void_p]
    lib.te_attn_rc_profile(buf)
    names = ["stage k", "barrier", "rows phase", "rows epilogue", "barrier+stage q", "cols phase", "cols epilogue"]
    for w in range(8):
        t = [buf[w * 16 + i] for i in range(8)]
        wall = (buf[w * 16 + 14] - buf[w * 16 + 15]) / 100.0      # 100 MHz constant counter -> us
        total = t[7] - t[0]
        print(f"wave {w}: total {total} cycles = {wall:.1f} us ({total / max(wall, 1e-9) / 1e3:.2f} GHz): " +
              ", ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)))


if __name__ == "__main__":
    main()
