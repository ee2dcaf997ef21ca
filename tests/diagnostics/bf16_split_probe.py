"""CPU diagnostic (test infrastructure; imports oracle/): would a SPLIT-OPERAND bf16 MFMA path be accurate enough for the
Linear.relprop rules?  -> profiles/r02_bf16_split_probe.log

Why ask: the step is bound by fp32 MFMA (157 TF peak, ~130-148 TF sustained); bf16 MFMA runs at 16x that rate.  An fp32
operand splits exactly into three bf16 parts (8 + 8 + 8 significand bits), a0 + a1 + a2, and a product of two bf16
values is exact in the fp32 accumulator of v_mfma_f32_*_bf16, so

    x3 : a0 b0 + a0 b1 + a1 b0                              (two-way split, error ~2^-16 per product)
    x6 : x3 + a0 b2 + a2 b0 + a1 b1                         (drops only terms <= 2^-24 |a||b|: fp32-class)
    x9 : all nine                                           (every product exact; only the fp32 accumulation rounds)

cost 3 / 6 / 9 bf16 MFMAs per fp32 one -- 5.3x / 2.7x / 1.8x the fp32-MFMA rate at equal efficiency.  This script
measures, on the CPU with bf16-exact fp32 tensors standing in for the MFMA operands, (1) the GEMM error of each scheme
against fp64 on the operands of a ViT-B block, next to a plain fp32 GEMM and a K-permuted fp32 GEMM, and (2) how far
the ViT-B relevance map moves when the three products of every Linear rule (|X||W|^T of the Z-pass, S W+ and S W- of
the C-pass) use a scheme, next to the map's own summation-order noise.
"""
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

torch.set_num_threads(8)
TERMS = {"x3": [(0, 0), (0, 1), (1, 0)],
         "x6": [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)],
         "x9": [(i, j) for i in range(3) for j in range(3)]}


def split3(x):
    """x (fp32) = p0 + p1 + p2 exactly, every part representable in bf16."""
    p0 = x.bfloat16().float()
    r = x - p0
    p1 = r.bfloat16().float()
    p2 = (r - p1).bfloat16().float()
    return p0, p1, p2


def mm_split(A, Bt, scheme):
    """A [.., K] x Bt [N, K]^T with the scheme's partial products, each an fp32-accumulated GEMM of bf16-exact operands
    (what one chain of bf16 MFMAs computes), summed smallest first."""
    a, b = split3(A), split3(Bt)
    parts = [a[i].matmul(b[j].t()) for i, j in sorted(TERMS[scheme], key=lambda t: -(t[0] + t[1]))]
    out = parts[0]
    for p in parts[1:]:
        out = out + p
    return out


def rule_with(scheme):
    """Linear.relprop as shipped (Z from the forward output) with its three products on `scheme`."""
    def rule(R, X, W, alpha=1.0, variant="ours"):
        pw, nw, px, nx = W.clamp(min=0), W.clamp(max=0), X.clamp(min=0), X.clamp(max=0)
        Yp = X.matmul(W.t())                                   # the forward output minus bias: a stock fp32 GEMM
        Z = 0.5 * (Yp + mm_split(X.abs(), W.abs(), scheme))
        S = O.safe_divide(R, Z)
        return alpha * (px * mm_split(S, pw.t().contiguous(), scheme) + nx * mm_split(S, nw.t().contiguous(), scheme))
    return rule


def ztrick_fp32(R, X, W, alpha=1.0, variant="ours"):
    pw, nw, px, nx = W.clamp(min=0), W.clamp(max=0), X.clamp(min=0), X.clamp(max=0)
    Z = 0.5 * (X.matmul(W.t()) + X.abs().matmul(W.abs().t()))
    S = O.safe_divide(R, Z)
    return alpha * (px * S.matmul(pw) + nx * S.matmul(nw))


def permk(R, X, W, alpha=1.0, variant="ours"):
    g = torch.Generator().manual_seed(5)
    p = torch.randperm(X.shape[-1], generator=g)
    return ztrick_fp32(R, X[..., p], W[:, p], alpha, variant)[..., torch.argsort(p)]


def gemm_errors(model, cache):
    print("GEMM error vs fp64 (max |err| / max |exact|, and RMS err / RMS exact):")
    c = cache["blocks"][5]
    cases = {"|X||W|^T  fc1 (K = 768)": (c["fc1_x"][0].abs(), c["fc1_w"].abs()),
             "|X||W|^T  fc2 (K = 3072)": (c["fc2_x"][0].abs(), c["fc2_w"].abs()),
             "mixed-sign X W^T  qkv": (c["qkv_x"][0], c["qkv_w"])}
    for name, (A, Bt) in cases.items():
        exact = A.double().matmul(Bt.double().t())
        g = torch.Generator().manual_seed(3)
        p = torch.randperm(A.shape[-1], generator=g)
        rows = {"fp32": A.matmul(Bt.t()), "fp32, K permuted": A[..., p].matmul(Bt[:, p].t())}
        for s in ("x3", "x6", "x9"):
            rows[s] = mm_split(A, Bt, s)
        line = []
        for k, v in rows.items():
            e = v.double() - exact
            line.append(f"{k}: {float(e.abs().max() / exact.abs().max()):.1e} / {float(e.pow(2).mean().sqrt() / exact.pow(2).mean().sqrt()):.1e}")
        print(f"  {name:28s} " + " | ".join(line), flush=True)


def main():
    model = vit.vit_base_patch16_224().eval()
    synthetic_init(model, 0)
    orig = O.linear_relprop
    first = True
    for seed, idxs in ((1, (0, 1)), (7, (2,))):
        x = seeded_randn((4 if seed == 7 else 2, 3, 224, 224), seed)
        for i in idxs:
            out = model(x[i:i + 1])
            oh = torch.zeros_like(out)
            oh.scatter_(1, out.argmax(-1, keepdim=True), 1.0)
            _attention_gradients((oh * out).sum(), [b.attn for b in model.blocks])
            cache = vit_cache_from_model(model)
            if first:
                gemm_errors(model, cache)
                print("ViT-B/16 map (start_layer 1): relative L-inf / min-max-normalised max deviation from the fp32 rule")
                first = False
            maps = {}
            for name, rule in (("fp32", ztrick_fp32), ("K-permuted fp32", permk), ("x3", rule_with("x3")),
                               ("x6", rule_with("x6")), ("x9", rule_with("x9"))):
                O.linear_relprop = rule
                try:
                    maps[name] = O.vit_relprop(oh.detach(), cache, 12, start_layer=1)["map"]
                finally:
                    O.linear_relprop = orig
            line = []
            for name in ("K-permuted fp32", "x3", "x6", "x9"):
                s = map_stats(maps[name], maps["fp32"])
                line.append(f"{name}: {s['rel_linf']:.1e} / {s['normalised_max_abs']:.1e}")
            print(f"  seed {seed} image {i}: " + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
