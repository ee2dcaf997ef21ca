#!/usr/bin/env python
"""Debug: the bench workload (ViT-B/16, batch DBG_B) eager vs HIP-graph replay: timing + equality, x6 on / off."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_harness import seeded_randn, synthetic_init  # noqa: E402
from transformer_explainability_amd import ops, vit  # noqa: E402
from transformer_explainability_amd.generators import LRP, GraphedCall  # noqa: E402

dev = torch.device("cuda:0")
model = vit.vit_base_patch16_224().eval()
synthetic_init(model, 0)
model.to(dev)
B = int(os.environ.get("DBG_B", "64"))
ops.USE_FUSED_PRODUCERS = os.environ.get("DBG_FUSED", "1") == "1"
x = seeded_randn((B, 3, 224, 224), 7).to(dev)
x2 = seeded_randn((B, 3, 224, 224), 8).to(dev)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for x6 in (True, False):
    ops.USE_LINEAR_X6 = x6
    lrp = LRP(model)
    f = lambda t: lrp.generate_LRP(t, method="transformer_attribution", start_layer=1)      # noqa: E731
    ms_e, e1 = timed(lambda: f(x))
    e1 = e1.clone()
    e2 = f(x2).clone()
    g = GraphedCall(f, (x,))
    ms_g, g1 = timed(lambda: g(x))
    g1 = g1.clone()
    g2 = g(x2).clone()
    print(f"B={B} x6={x6} fused={ops.USE_FUSED_PRODUCERS}: eager {ms_e:.1f} ms, graph {ms_g:.1f} ms; graph(x)==eager(x) "
          f"{bool(torch.equal(g1, e1))}, graph(x2)==eager(x2) {bool(torch.equal(g2, e2))}, max diff "
          f"{float((g2 - e2).abs().max()):.3e} (max {float(e2.abs().max()):.3e})", flush=True)
    del g
