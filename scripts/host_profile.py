#!/usr/bin/env python
"""Diagnostic: host-side enqueue time of one generate_LRP step by phase (no device sync between phases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__
from transformer_explainability_amd import ops, vit
from transformer_explainability_amd.generators import LRP, _one_hot, _attention_gradients

__graft_entry__.build()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = vit.vit_base_patch16_224().eval().to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 3, 224, 224, device=dev)
lrp = LRP(model)
for _ in range(2):
    lrp.generate_LRP(x, start_layer=1)
torch.cuda.synchronize()
acc = {"forward": 0.0, "onehot+backward": 0.0, "relprop": 0.0}
n = 4
calls = {}
orig = {}
for name in ("linear_relprop", "matmul_relprop_av", "matmul_relprop_qk", "add_relprop", "clone_relprop", "gradcam_headmean", "rollout", "index_select_relprop"):
    f = getattr(ops, name)
    orig[name] = f
    def mk(f, name):
        def w(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); calls[name] = calls.get(name, 0.0) + time.perf_counter() - t; return r
        return w
    setattr(ops, name, mk(f, name))
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model(x)
    t1 = time.perf_counter()
    oh = _one_hot(out, None)
    loss = torch.sum(oh * out)
    _attention_gradients(loss, [blk.attn for blk in model.blocks])
    t2 = time.perf_counter()
    model.relprop(oh, method="transformer_attribution", start_layer=1, alpha=1)
    t3 = time.perf_counter()
    acc["forward"] += t1 - t0; acc["onehot+backward"] += t2 - t1; acc["relprop"] += t3 - t2
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"step: enqueue {1e3*(t3-t0):.1f} ms, total with sync {1e3*(t4-t0):.1f} ms")
for k, v in acc.items():
    print(f"host enqueue {k:16s} {1e3*v/n:7.2f} ms/step")
for k, v in sorted(calls.items(), key=lambda kv: -kv[1]):
    print(f"   ops.{k:22s} {1e3*v/n:7.2f} ms/step")
