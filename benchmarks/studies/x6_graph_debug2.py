#!/usr/bin/env python
"""Debug: GraphedLRP replay on NEW inputs vs eager, per block, with the Linear rules on x6 / fp32-MFMA."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.ref_harness import seeded_randn, synthetic_init  # noqa: E402
from transformer_explainability_amd import ops, vit  # noqa: E402
from transformer_explainability_amd.generators import LRP, GraphedLRP  # noqa: E402

dev = torch.device("cuda:0")
model = vit.vit_base_patch16_224().eval()
synthetic_init(model, 0)
model.to(dev)
B = int(os.environ.get("DBG_B", "4"))
x = seeded_randn((B, 3, 224, 224), 7).to(dev)
x2 = seeded_randn((B, 3, 224, 224), 8).to(dev)
for x6 in (True, False):
    ops.USE_LINEAR_X6 = x6
    lrp = LRP(model)
    base = lrp.generate_LRP(x, start_layer=1).clone()
    glrp = GraphedLRP(lrp, x, method="transformer_attribution", start_layer=1)
    r1 = glrp(x).clone()
    print(f"x6={x6}: replay(x) == eager(x): {bool(torch.equal(r1, base))}")
    r2 = glrp(x2).clone()
    cams_g = [blk.attn.get_attn_cam().clone() for blk in model.blocks]
    logits_g = model.head.Y.detach().clone()
    e2 = lrp.generate_LRP(x2, start_layer=1).clone()
    cams_e = [blk.attn.get_attn_cam().clone() for blk in model.blocks]
    print(f"x6={x6}: replay(x2) == eager(x2): {bool(torch.equal(r2, e2))}; logits equal {bool(torch.equal(logits_g, model.head.Y.detach()))}")
    for l in reversed(range(12)):
        d = float((cams_g[l] - cams_e[l]).abs().max())
        print(f"   block {l}: attn_cam max diff {d:.3e} (max {float(cams_e[l].abs().max()):.3e})")
    r3 = glrp(x).clone()
    print(f"x6={x6}: replay(x) again == eager(x): {bool(torch.equal(r3, base))}")
    del glrp
