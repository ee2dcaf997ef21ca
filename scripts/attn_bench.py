#!/usr/bin/env python
"""GPU measurement: the two attention relprop rules at the bench shapes (ViT-B/16, batch 64), HIP-event time per rule."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import ops  # noqa: E402


def t(fn, reps=10):
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
    B, H, N, D = (int(a) for a in (sys.argv[1:5] + [64, 12, 197, 64][len(sys.argv) - 1:]))
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, device=d) for _ in range(3))
    zqk = q @ k.transpose(-1, -2)
    attn = torch.softmax(zqk * D ** -0.5, -1)
    zav = attn @ v
    R = torch.randn(B, H, N, D, device=d) * 0.01
    Rnn = torch.randn(B, H, N, N, device=d) * 0.01
    av = t(lambda: ops.matmul_relprop_av(R, attn, v, out_scale=0.5, z=zav))
    qk = t(lambda: ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk))
    if len(sys.argv) > 5 and sys.argv[5] == "producers" and ops.attention_forward_supported(N, D):
        qkv = torch.randn(B, N, 3 * H * D, device=d)
        g = torch.randn(B, N, H * D, device=d)
        out_p, attn_p, _ = ops.attention_forward(qkv, H, D ** -0.5)
        fw = t(lambda: ops.attention_forward(qkv, H, D ** -0.5))
        bw = t(lambda: ops.attention_backward(g, qkv, attn_p, H, D ** -0.5, out=out_p))
        print(f"B={B} H={H} N={N}: producer forward {fw:7.1f} us   backward {bw:7.1f} us")
    nn = B * H * N * N * 4 / 1e6
    print(f"B={B} H={H} N={N}: AV rule {av:7.1f} us   QK rule {qk:7.1f} us   (one [B,H,N,N] tensor = {nn:.0f} MB)")


if __name__ == "__main__":
    main()
