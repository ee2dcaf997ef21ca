#!/usr/bin/env python
"""GPU diagnostic: where do a batch-of-B run and B batch-1 runs of generate_LRP diverge?

Records the output of every C-ABI op of both runs and prints the first ops whose per-sample slices
differ, then runs the CPU oracle on the cached tensors of the worst sample (batch and single) to tell a
kernel bug (HIP != oracle on the same cache) from input sensitivity (oracle moves just as much).
Test/diagnostic infrastructure: imports oracle/, never used by the product path.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import relprop_oracle as O  # noqa: E402
from oracle.model_cache import vit_cache_from_model  # noqa: E402
from oracle.ref_harness import seeded_randn, synthetic_init  # noqa: E402
from transformer_explainability_amd import ops, vit  # noqa: E402
from transformer_explainability_amd.generators import LRP  # noqa: E402

NAMES = ["linear_relprop", "matmul_relprop_av", "matmul_relprop_qk", "add_relprop", "clone_relprop",
         "index_select_relprop", "gradcam_headmean", "rollout"]
LOG = []


def wrap(name):
    fn = getattr(ops, name)

    def w(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else [out]
        LOG.append((name, [o.detach().clone() for o in outs]))
        return out
    return w


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    sl = 1
    d = torch.device("cuda:0")
    model = vit.vit_base_patch16_224().eval()
    synthetic_init(model, 0)
    model.to(d)
    x = seeded_randn((B, 3, 224, 224), seed).to(d)
    lrp = LRP(model)
    for n in NAMES:
        setattr(ops, n, wrap(n))

    LOG.clear()
    mb = lrp.generate_LRP(x, start_layer=sl).clone()
    log_b = list(LOG)
    logits_b = model.head.Y.detach().clone()
    cache_b = vit_cache_from_model(model)
    worst = (0.0, 0)
    caches_s = []
    for i in range(B):
        LOG.clear()
        ms = lrp.generate_LRP(x[i:i + 1], start_layer=sl).clone()
        log_s = list(LOG)
        caches_s.append(vit_cache_from_model(model))
        lg = model.head.Y.detach()
        print(f"sample {i}: argmax batch {int(logits_b[i].argmax())} single {int(lg[0].argmax())} "
              f"logit maxdiff {float((logits_b[i] - lg[0]).abs().max()):.3g}")
        rel = float((mb[i] - ms[0]).abs().max() / ms.abs().max())
        print(f"  map rel diff {rel:.3g}  max|single| {float(ms.abs().max()):.3g}")
        if rel > worst[0]:
            worst = (rel, i)
        shown = 0
        for j, ((nb, ob), (ns, os_)) in enumerate(zip(log_b, log_s)):
            assert nb == ns
            for t, (tb, ts) in enumerate(zip(ob, os_)):
                tb_i = tb[i:i + 1]
                if tb_i.shape != ts.shape:
                    continue
                dd = float((tb_i - ts).abs().max())
                mx = float(ts.abs().max())
                r = dd / max(mx, 1e-30)
                if r > 1e-3 and shown < 12:
                    shown += 1
                    sgn = float((torch.sign(tb_i) != torch.sign(ts)).float().mean())
                    print(f"    op#{j} {nb}[{t}] shape {tuple(ts.shape)} rel {r:.3g} max {mx:.3g} sign-flip frac {sgn:.3g}")
    # oracle on the worst sample's caches
    rel, i = worst
    print(f"worst sample {i} rel {rel:.3g}")

    def sub(cache, i):
        def s(t):
            return t[i:i + 1] if (torch.is_tensor(t) and t.dim() >= 2 and t.shape[0] == B and t.shape[0] != t.shape[-1]) else t
        out = {k: s(v) for k, v in cache.items() if k != "blocks"}
        out["blocks"] = [{k: (v if (k.endswith("_w") or v is None) else v[i:i + 1]) for k, v in blk.items()}
                         for blk in cache["blocks"]]
        return out
    for tag, cache in (("batch-cache", sub(cache_b, i)), ("single-cache", caches_s[i])):
        lg = logits_b[i:i + 1].float().cpu()
        oh = torch.zeros_like(lg)
        oh.scatter_(1, lg.argmax(-1, keepdim=True), 1.0)
        for dt in (torch.float32, torch.float64):
            c = {k: (v.to(dt) if torch.is_tensor(v) else v) for k, v in cache.items() if k != "blocks"}
            c["blocks"] = [{k: (None if v is None else v.to(dt)) for k, v in blk.items()} for blk in cache["blocks"]]
            ref = O.vit_relprop(oh.to(dt), c, num_heads=12, start_layer=sl)["map"].float()
            print(f"  oracle[{tag},{dt}] vs HIP batch: {float((ref[0] - mb[i].cpu()).abs().max() / ref.abs().max()):.3g}"
                  f"  max|ref| {float(ref.abs().max()):.3g}")


if __name__ == "__main__":
    main()
