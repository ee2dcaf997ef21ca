"""CPU diagnostic (test infrastructure; imports oracle/): how far does the map move when Linear.relprop's Z is derived
from the forward output, vs a K-permutation of the two-product form, vs fp64?  -> profiles/r01_zpass_from_forward_probe.log"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import map_stats
from oracle import relprop_oracle as O
from oracle.model_cache import vit_cache_from_model
from oracle.ref_harness import seeded_randn, synthetic_init
from transformer_explainability_amd import vit
from transformer_explainability_amd.generators import _attention_gradients
torch.set_num_threads(8)
model = vit.vit_base_patch16_224().eval(); synthetic_init(model, 0)
orig = O.linear_relprop
def ztrick(R, X, W, alpha=1.0, variant="ours"):
    pw, nw, px, nx = W.clamp(min=0), W.clamp(max=0), X.clamp(min=0), X.clamp(max=0)
    Yp = X.matmul(W.t())                       # forward output minus bias (fp32 GEMM)
    A = X.abs().matmul(W.abs().t())
    Z = 0.5 * (Yp + A)
    S = O.safe_divide(R, Z)
    return alpha * (px * S.matmul(pw) + nx * S.matmul(nw))
def permk(R, X, W, alpha=1.0, variant="ours"):
    g = torch.Generator().manual_seed(5); p = torch.randperm(X.shape[-1], generator=g)
    return orig(R, X[..., p], W[:, p], alpha, variant)[..., torch.argsort(p)]
for seed, idxs in ((1,(0,1)),(7,(2,3))):
    x = seeded_randn((4 if seed==7 else 2, 3, 224, 224), seed)
    for i in idxs:
        out = model(x[i:i+1]); oh = torch.zeros_like(out); oh.scatter_(1, out.argmax(-1, keepdim=True), 1.0)
        _attention_gradients((oh*out).sum(), [b.attn for b in model.blocks])
        cache = vit_cache_from_model(model)
        for sl in (0,1):
            O.linear_relprop = orig
            ref = O.vit_relprop(oh.detach(), cache, 12, start_layer=sl)["map"]
            c64 = {k:(v.double() if torch.is_tensor(v) else v) for k,v in cache.items() if k!="blocks"}; c64["blocks"]=[{k:v.double() for k,v in b.items()} for b in cache["blocks"]]
            ref64 = O.vit_relprop(oh.detach().double(), c64, 12, start_layer=sl)["map"].float()
            O.linear_relprop = ztrick
            alt = O.vit_relprop(oh.detach(), cache, 12, start_layer=sl)["map"]
            O.linear_relprop = permk
            prm = O.vit_relprop(oh.detach(), cache, 12, start_layer=sl)["map"]
            O.linear_relprop = orig
            s1 = map_stats(alt, ref); s2 = map_stats(prm, ref); s3 = map_stats(ref, ref64); s4 = map_stats(alt, ref64)
            print(f"seed {seed} sample {i} sl {sl}: ztrick vs ref rel {s1['rel_linf']:.2e} norm {s1['normalised_max_abs']:.2e} | K-permuted ref vs ref rel {s2['rel_linf']:.2e} | ref vs fp64 rel {s3['rel_linf']:.2e} | ztrick vs fp64 {s4['rel_linf']:.2e}", flush=True)
