#!/usr/bin/env python
"""Are the attention rule kernels bitwise repeatable under concurrent load?  Runs each rule 30 times on the same inputs while a
second stream keeps the CUs busy with GEMMs, and compares every result with the first."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import ops  # noqa: E402
d = torch.device("cuda:0")
B, H, N, D = 64, 12, 197, 64
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, device=d) for _ in range(3))
zqk = q @ k.transpose(-1, -2)
attn = torch.softmax(zqk * D ** -0.5, -1)
zav = attn @ v
R = torch.randn(B, H, N, D, device=d) * 0.01
Rnn = torch.randn(B, H, N, N, device=d) * 0.01
qkv = torch.randn(B, N, 3 * H * D, device=d)
g = torch.randn(B, N, H * D, device=d)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=d)
MODE = sys.argv[1] if len(sys.argv) > 1 else "gemm"
_, attn_side, _ = ops.attention_forward(qkv, H, D ** -0.5)
def load():
    with torch.cuda.stream(side):
        if MODE == "gemm":
            for _ in range(4):
                torch.mm(a, a)
        elif MODE == "attnbwd":
            for _ in range(2):
                ops.attention_backward(g, qkv, attn_side, H, D ** -0.5)
        else:
            for _ in range(2):
                ops.matmul_relprop_av(R, attn, v, out_scale=0.5, z=zav)
                ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk)
def rep(name, fn, n=40):
    first = [t.clone() for t in fn()]
    bad = 0
    for i in range(n):
        load()
        out = fn()
        torch.cuda.synchronize()
        if not all(torch.equal(x, y) for x, y in zip(out, first)):
            bad += 1
            diffs = [float((x - y).abs().max()) for x, y in zip(out, first)]
            if bad <= 3:
                print(f"  {name}: run {i} differs, max abs diff per output {diffs}", flush=True)
    print(f"{name}: {bad} of {n} runs differ from the first", flush=True)
rep("AV rule", lambda: ops.matmul_relprop_av(R, attn, v, out_scale=0.5, z=zav))
rep("QK rule", lambda: ops.matmul_relprop_qk(Rnn, q, k, out_scale=0.5, z=zqk))
if ops.attention_forward_supported(N, D):
    _, attn_p, _ = ops.attention_forward(qkv, H, D ** -0.5)
    rep("attention forward", lambda: ops.attention_forward(qkv, H, D ** -0.5))
    rep("attention backward", lambda: ops.attention_backward(g, qkv, attn_p, H, D ** -0.5))
