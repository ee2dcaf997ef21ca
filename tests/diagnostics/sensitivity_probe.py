#!/usr/bin/env python
"""CPU probe (test infrastructure, uses oracle/): how far does the fp32 relevance map of a random-init ViT-B/16
move when the PRODUCERS (forward + attention-gradient backward) are perturbed at fp32-rounding level?

The relprop chain divides by near-zero mixed-sign sums (q.k^T, x0+x1), so it amplifies producer noise; this
quantifies the band that any comparison against maps produced on a different machine / BLAS has to live with
(DESIGN.md section 4).  Perturbation: every cached producer tensor is multiplied by (1 + eps * U(-1,1)),
eps = 2^-23 (one ulp-level relative noise), the relprop itself (the oracle) is unchanged."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import map_stats  # noqa: E402
from oracle import relprop_oracle as O  # noqa: E402
from oracle.model_cache import vit_cache_from_model  # noqa: E402
from oracle.ref_harness import seeded_randn, synthetic_init  # noqa: E402
from transformer_explainability_amd import vit  # noqa: E402
from transformer_explainability_amd.generators import _attention_gradients  # noqa: E402


def perturb(cache, eps, gen):
    def p(t):
        return None if t is None else t * (1 + eps * (2 * torch.rand(t.shape, generator=gen) - 1))
    out = {k: (v if k.endswith("_w") else p(v)) for k, v in cache.items() if k != "blocks"}
    out["blocks"] = [{k: (v if k.endswith("_w") else p(v)) for k, v in b.items()} for b in cache["blocks"]]
    return out


def main():
    torch.set_num_threads(8)
    model = vit.vit_base_patch16_224().eval()
    synthetic_init(model, 0)
    for seed, n in ((1, 2), (7, 4)):
        x = seeded_randn((n, 3, 224, 224), seed)
        for i in range(n):
            out = model(x[i:i + 1])
            oh = torch.zeros_like(out)
            oh.scatter_(1, out.argmax(-1, keepdim=True), 1.0)
            _attention_gradients((oh * out).sum(), [b.attn for b in model.blocks])
            cache = vit_cache_from_model(model)
            for sl in (0, 1):
                ref = O.vit_relprop(oh.detach(), cache, 12, start_layer=sl)["map"]
                worst = {"raw_max_abs": 0, "normalised_max_abs": 0, "rel_linf": 0}
                for trial in range(3):
                    g = torch.Generator().manual_seed(100 + trial)
                    alt = O.vit_relprop(oh.detach(), perturb(cache, 2.0 ** -23, g), 12, start_layer=sl)["map"]
                    s = map_stats(alt, ref)
                    worst = {k: max(worst[k], s[k]) for k in worst}
                print(f"seed {seed} sample {i} start_layer {sl}: max|map| {float(ref.abs().max()):.3g}  "
                      f"1-ulp producer noise moves it by raw {worst['raw_max_abs']:.3g} "
                      f"normalised {worst['normalised_max_abs']:.3g} relative {worst['rel_linf']:.3g}", flush=True)


if __name__ == "__main__":
    main()
