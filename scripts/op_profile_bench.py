#!/usr/bin/env python
"""Which stock (ATen) kernels remain in the BENCH configuration's step (fused producers, x6 Linear layers) and where they are
called from: torch profiler with stacks over one eager step, the aten ops by self device time with the innermost frame of this
package."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import ops, vit  # noqa: E402
from transformer_explainability_amd.generators import LRP  # noqa: E402

d = torch.device("cuda:0")
torch.manual_seed(0)
ops.USE_FUSED_PRODUCERS = True
ops.X6_GEMM = "auto"
ops.USE_LINEAR_X6 = True
ops.X6_TILE = 2
model = vit.vit_base_patch16_224().eval().to(d)
x = torch.randn(64, 3, 224, 224, device=d)
lrp = LRP(model)
for _ in range(2):
    lrp.generate_LRP(x, start_layer=1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    lrp.generate_LRP(x, start_layer=1)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    frame = next((s for s in ev.stack if "transformer-explainability_amd" in s or "transformer_explainability_amd" in s), "(autograd / no package frame)")
    key = (ev.name, str(ev.input_shapes)[:60], frame.split("transformer")[-1][:70])
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += ev.self_device_time_total
for (name, shp, frame), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{name:28s} {n:4d} calls {t:9.1f} us  {shp:60s} {frame}")
