#!/usr/bin/env python
"""GPU measurement: at N <= 224 (the ViT-B/16 shapes), the one-workgroup-per-head producers (te_attn_fwd6 / av6_kb + qk_rc) against the
chunked long-sequence producers (te_attn_fwd6l / te_attn_bwd6l) run on the same fused qkv activation through the strided entry points."""
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


d = torch.device("cuda:0")
for B, H, N in ((64, 12, 197), (64, 12, 224), (32, 12, 128)):
    D, C = 64, H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * C, device=d)
    g = torch.randn(B, N, C, device=d)
    sc = D ** -0.5
    out, attn, _ = ops.attention_forward(qkv, H, sc)
    th = lambda x, i: x[..., i * C:(i + 1) * C]      # noqa: E731
    f_s = t(lambda: ops.attention_forward(qkv, H, sc))
    f_l = t(lambda: ops.attention_forward_qkv(th(qkv, 0), th(qkv, 1), th(qkv, 2), H, sc))
    b_s = t(lambda: ops.attention_backward(g, qkv, attn, H, sc, out=out))
    dq = torch.empty_like(qkv)
    b_l = t(lambda: ops.attention_backward_qkv(g, th(qkv, 0), th(qkv, 1), th(qkv, 2), attn, H, sc, th(dq, 0), th(dq, 1), th(dq, 2), out=out))
    o2, a2, _, _ = ops.attention_forward_qkv(th(qkv, 0), th(qkv, 1), th(qkv, 2), H, sc)
    print(f"B={B} H={H} N={N}: forward short {f_s:7.1f} us  long-structure {f_l:7.1f} us   backward short {b_s:7.1f} us  long-structure {b_l:7.1f} us"
          f"   (max |d attn| {float((a2 - attn).abs().max()):.2e})")
