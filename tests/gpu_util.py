"""Helpers shared by the `-m gpu` parity tests (HIP path through the C ABI vs the CPU oracle)."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def stats(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    d = (got - ref).abs()
    return {"max_abs": float(d.max()), "ref_max": float(ref.abs().max()),
            "rel": float(d.max()) / max(float(ref.abs().max()), 1e-30),
            "nonfinite": int((~torch.isfinite(got)).sum())}


def record(name, **kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps({"test": name, **kw}) + "\n")


def check(name, got, ref, rel_tol, exact=False):
    s = stats(got, ref)
    s["bitwise_equal"] = bool(torch.equal(got.detach().cpu(), ref.detach().cpu()))
    record(name, **s, rel_tol=rel_tol)
    assert s["nonfinite"] == 0, (name, s)
    if exact:
        assert s["bitwise_equal"], (name, s)
    assert s["rel"] <= rel_tol, (name, s)
    return s


from oracle.model_cache import bert_cache_from_model, vit_cache_from_model  # noqa: E402,F401


def minmax(m):
    flat = m.reshape(m.shape[0], -1)
    lo, hi = flat.min(1, keepdim=True).values, flat.max(1, keepdim=True).values
    return (flat - lo) / (hi - lo)


def map_stats(got, ref):
    """The three parity statistics of SURVEY.md section 8d: raw, min-max-normalised, relative."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    raw = float((got - ref).abs().max())
    return {"raw_max_abs": raw, "normalised_max_abs": float((minmax(got) - minmax(ref)).abs().max()),
            "rel_linf": raw / max(float(ref.abs().max()), 1e-30), "ref_max": float(ref.abs().max())}
