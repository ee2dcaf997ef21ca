#!/usr/bin/env python
"""GPU measurement: nn.GELU as a producer of operand planes (te_gelu_{forward,backward}_x6_planes_f32) against the passes it
replaces (te_gelu_*_f32 followed by te_linear_x6_split_*), HIP-event time per call at the Mlp shape of the bench
(ViT-B/16 batch 64: 12608 x 3072)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import _lib, ops  # noqa: E402


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    d = torch.device("cuda:0")
    T, K = (int(a) for a in (sys.argv[1:3] + [12608, 3072][len(sys.argv) - 1:]))
    lib = _lib.load()
    torch.manual_seed(0)
    x = torch.randn(T, K, device=d) * 2
    dy = torch.randn(T, K, device=d)
    nb = lib.te_linear_x6_planes_bytes(T, K)
    p0, p1 = (torch.empty(nb, dtype=torch.uint8, device=d) for _ in range(2))
    s = torch.cuda.current_stream().cuda_stream
    y = ops.gelu_forward(x)
    dx = ops.gelu_backward(dy, x)
    n = T * K
    rows = [
        ("gelu_backward (fp32 out)", lambda: ops.gelu_backward(dy, x), 12),
        ("split_matrix of d_h", lambda: lib.te_linear_x6_split_matrix_f32(dx.data_ptr(), T, K, 0, p0.data_ptr(), nb, s), 10),
        ("gelu_backward -> planes", lambda: ops.gelu_backward_planes(dy, x), 14),
        ("gelu_forward (fp32 out)", lambda: ops.gelu_forward(x), 8),
        ("split_dual of a", lambda: lib.te_linear_x6_split_dual_f32(y.data_ptr(), T, K, p0.data_ptr(), p1.data_ptr(), nb, s), 16),
        ("gelu_forward -> fp32 + planes", lambda: ops.gelu_forward_planes(x), 20),
    ]
    print(f"T={T} K={K} TE_GELU_SPLIT={os.environ.get('TE_GELU_SPLIT', '(default: staged)')}")
    for name, fn, bpe in rows:
        us = t(fn)
        print(f"  {name:32s} {us:8.1f} us   {bpe * n / us / 1e6:6.2f} TB/s ({bpe} B per element)")


if __name__ == "__main__":
    main()
