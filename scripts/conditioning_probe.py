"""CPU probe: can a synthetic ViT-B/16 be initialised so that the REFERENCE reproduces its own
transformer_attribution map to 1e-4 (min-max normalised) under rounding-level noise?  Runs the reference's generate_LRP on
three images per variant with 6 draws of make_golden._RoundingNoise.  Result: profiles/r04_conditioning_probe_cpu.log (no).

    python scripts/conditioning_probe.py [default qk.3v.5s1 v2 s3 s3v2]      (needs the reference checkout or its stage)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, numpy as np
import make_golden as G
from oracle import ref_harness as rh

def conditioned_init(model, seed, qk_bias=0.3, v_bias=0.5, stream=1.0, wscale=1.0):
    rh.synthetic_init(model, seed)
    with torch.no_grad():
        C = model.blocks[0].attn.qkv.weight.shape[1]
        for blk in model.blocks:
            b = blk.attn.qkv.bias
            b[:2 * C] += qk_bias
            b[2 * C:] += v_bias
            if wscale != 1.0:
                blk.attn.qkv.weight[:2 * C].mul_(wscale)
        model.pos_embed += stream

vit = rh.load_reference_vit()
torch.set_num_threads(8)
variants = {"v2": dict(qk_bias=0.5, v_bias=2.0, stream=3.0, wscale=0.25), "s3": dict(qk_bias=0.0, v_bias=0.0, stream=3.0),
            "s3v2": dict(qk_bias=0.0, v_bias=2.0, stream=3.0), "default": None, "v0.5": dict(qk_bias=0.0, v_bias=0.5, stream=0.0), "qk.3v.5": dict(qk_bias=0.3, v_bias=0.5, stream=0.0),
            "qk.3v.5s1": dict(qk_bias=0.3, v_bias=0.5, stream=1.0)}
only = sys.argv[1:] or list(variants)
for name in only:
    kw = variants[name]
    model = vit["ViT_LRP"].vit_base_patch16_224(pretrained=False).eval()
    if kw is None: rh.synthetic_init(model, 0)
    else: conditioned_init(model, 0, **kw)
    g32 = vit["gen"].LRP(model)
    xs = rh.seeded_randn((3, 3, 224, 224), 21)
    for i in range(3):
        x = xs[i:i+1]
        for sl in (0, 1):
            t0 = time.time()
            base = g32.generate_LRP(x, method="transformer_attribution", start_layer=sl).detach().clone()
            ds = []
            for dr in range(6):
                with G._RoundingNoise(model, dr):
                    m = g32.generate_LRP(G._ulp_noise(x, dr), method="transformer_attribution", start_layer=sl).detach().clone()
                ds.append(G._dist(m, base)[0])
            print(f"{name} img{i} sl{sl}: range {float(base.max()-base.min()):.2e} draws median {np.median(ds):.1e} max {max(ds):.1e}  ({time.time()-t0:.0f}s)", flush=True)
