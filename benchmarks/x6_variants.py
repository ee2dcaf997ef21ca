#!/usr/bin/env python
"""Round-4 A/B of the x6 Linear kernels on the ViT-B/16 batch-64 shapes (T = 12 608): per shape the Z-pass, the C-pass
(phases of te_linear_relprop_x6_f32) and the plain product in both directions (te_gemm_x6_f32), under flag variants of
the SAME build -- two vs three LDS stages (TE_X6_STAGES_3), tile geometry per pass.  HIP-event medians over --iters
launches, variants interleaved per shape so that box-to-box and thermal drift cancel; results are bit-identical across
variants (asserted).

    python benchmarks/x6_variants.py [--iters 12] [--variants base,st3,z128,c128,g64] [--config vit_b16]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from transformer_explainability_amd import _lib, ops  # noqa: E402

CONFIGS = {
    "vit_b16": (64 * 197, [("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]),
    "vit_l16": (32 * 577, [("qkv", 1024, 3072), ("proj", 1024, 1024), ("fc1", 1024, 4096), ("fc2", 4096, 1024)]),
    "bert_base": (32 * 512, [("qkv1", 768, 768), ("inter", 768, 3072), ("out", 3072, 768)]),
}
Z128, Z256 = 1 << ops.TE_X6_TILE_Z_SHIFT, 2 << ops.TE_X6_TILE_Z_SHIFT
C128, C256 = 1 << ops.TE_X6_TILE_C_SHIFT, 2 << ops.TE_X6_TILE_C_SHIFT
VARIANTS = {
    "base": 0,                            # default policy: two LDS stages, tile geometry chosen per launch
    "st3": ops.TE_X6_STAGES_3,            # three stages where the tile is 256 weight rows
    "z128": Z128,                         # Z-pass on 128-row tiles (two workgroups per CU: epilogue beside the other's loop)
    "z256": Z256,
    "c128": C128,
    "c256": C256,
    "g128": 1,                            # plain products on 128-row tiles (both passes too)
    "g256": 2,
    "g64": 3,                             # every launch on 128 x 128 tiles (three workgroups per CU)
    "z64": 3 << ops.TE_X6_TILE_Z_SHIFT,
    "c64": 3 << ops.TE_X6_TILE_C_SHIFT,
    # round 6, study builds only (TE_RELPROP_LIB=.../libte_relprop_study.so): schedule options of x6_kernel, set through TE_X6_OPT
    # "opt<bits>" (any value): default flags, TE_X6_OPT=<bits> while the variant runs (kX6Opt*: 4 = no epilogue, 8 = warm-up requests)
}


def med(evs):
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--variants", default="base,g128,g64")
    ap.add_argument("--config", default="vit_b16")
    a = ap.parse_args()
    _lib.require_device()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    names = a.variants.split(",")
    T, shapes = CONFIGS[a.config]
    totals = {v: dict(z=0.0, c=0.0, fwd=0.0, bwd=0.0) for v in names}
    for lname, in_f, out_f in shapes:
        X = torch.randn(T, in_f, generator=g).to(dev)
        W = (0.03 * torch.randn(out_f, in_f, generator=g)).to(dev)
        b = (0.1 * torch.randn(out_f, generator=g)).to(dev)
        R = (1e-3 * torch.randn(T, out_f, generator=g)).to(dev)
        dY = torch.randn(T, out_f, generator=g).to(dev)
        Y = torch.nn.functional.linear(X, W, b)
        cache = {}
        planes = ops.x6_weight_planes(W, cache)
        wp_f, wp_b = ops.x6_matrix_planes(W, False, cache), ops.x6_matrix_planes(W, True, cache)
        nb = lib.te_linear_x6_planes_bytes
        xs, xa = torch.empty(nb(T, in_f), dtype=torch.uint8, device=dev), torch.empty(nb(T, in_f), dtype=torch.uint8, device=dev)
        dys = torch.empty(nb(T, out_f), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.te_linear_x6_split_dual_f32(X.data_ptr(), T, in_f, xs.data_ptr(), xa.data_ptr(), xs.numel(), st), "split")
        _lib.check(lib.te_linear_x6_split_matrix_f32(dY.data_ptr(), T, out_f, 0, dys.data_ptr(), dys.numel(), st), "split")
        ws = torch.empty(lib.te_linear_relprop_x6_workspace_bytes(T, in_f, out_f), dtype=torch.uint8, device=dev)
        wsg = torch.empty(max(lib.te_gemm_x6_workspace_bytes(T, in_f, out_f), lib.te_gemm_x6_workspace_bytes(T, out_f, in_f)),
                          dtype=torch.uint8, device=dev)
        out = torch.empty(T, in_f, device=dev)
        yo, dxo = torch.empty(T, out_f, device=dev), torch.empty(T, in_f, device=dev)
        status = ops.x6_status(dev)

        def rule(flags):
            _lib.check(lib.te_linear_relprop_x6_f32(R.data_ptr(), None, 0, 1, X.data_ptr(), W.data_ptr(), planes.data_ptr(),
                                                    xa.data_ptr(), Y.data_ptr(), b.data_ptr(), out.data_ptr(), T, in_f, out_f,
                                                    flags, status.data_ptr(), ws.data_ptr(), ws.numel(), st), "rule")

        def fwd(flags):
            _lib.check(lib.te_gemm_x6_f32(X.data_ptr(), xs.data_ptr(), wp_f.data_ptr(), b.data_ptr(), yo.data_ptr(), T, in_f,
                                          out_f, flags & ~0x3c00, status.data_ptr(), wsg.data_ptr(), wsg.numel(), st), "fwd")

        def bwd(flags):
            _lib.check(lib.te_gemm_x6_f32(dY.data_ptr(), dys.data_ptr(), wp_b.data_ptr(), None, dxo.data_ptr(), T, out_f,
                                          in_f, flags & ~0x3c00, status.data_ptr(), wsg.data_ptr(), wsg.numel(), st), "bwd")

        ev = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
        rec = {v: dict(z=[], c=[], fwd=[], bwd=[]) for v in names}
        ref = {}
        for it in range(a.iters + 2):
            for v in names:
                fl = 0 if v.startswith("opt") else VARIANTS[v]
                if v.startswith("opt"):
                    os.environ["TE_X6_OPT"] = v[3:]
                else:
                    os.environ.pop("TE_X6_OPT", None)
                pairs = {}
                rule(fl | ops.TE_X6_PHASE_SPLIT)
                for key, fn in (("z", lambda: rule(fl | ops.TE_X6_PHASE_Z)), ("c", lambda: rule(fl | ops.TE_X6_PHASE_C)),
                                ("fwd", lambda: fwd(fl)), ("bwd", lambda: bwd(fl))):
                    s, e = ev(), ev()
                    s.record()
                    fn()
                    e.record()
                    pairs[key] = (s, e)
                if it >= 2:
                    for key, pr in pairs.items():
                        rec[v][key].append(pr)
                if it == 0:
                    torch.cuda.synchronize()
                    cur = (out.clone(), yo.clone(), dxo.clone())
                    if not ref:
                        ref["r"] = cur
                    else:
                        garbage = v.startswith("opt") and int(v[3:]) & 0x74      # ablations of the epilogue: wrong by construction
                        assert garbage or all(torch.equal(x, y) for x, y in zip(cur, ref["r"])), f"{v}: results differ from {names[0]}"
        torch.cuda.synchronize()
        ops.x6_raise_if_failed(dev)
        gemm = 2.0 * T * in_f * out_f
        for v in names:
            r = {k: med(rec[v][k]) for k in ("z", "c", "fwd", "bwd")}
            for k in r:
                totals[v][k] += r[k]
            print("VAR " + json.dumps(dict(config=a.config, layer=lname, variant=v, in_f=in_f, out_f=out_f,
                                           z_us=round(r["z"], 1), c_us=round(r["c"], 1), fwd_us=round(r["fwd"], 1),
                                           bwd_us=round(r["bwd"], 1),
                                           z_frac=round(6 * gemm / r["z"] * 1e-6 / 2500, 3),
                                           c_frac=round(12 * gemm / r["c"] * 1e-6 / 2500, 3),
                                           fwd_frac=round(6 * gemm / r["fwd"] * 1e-6 / 2500, 3),
                                           bwd_frac=round(6 * gemm / r["bwd"] * 1e-6 / 2500, 3))), flush=True)
    for v in names:
        t = totals[v]
        print(f"TOTAL {a.config} {v}: Z {t['z']:.0f} C {t['c']:.0f} fwd {t['fwd']:.0f} bwd {t['bwd']:.0f} us  "
              f"sum {sum(t.values()):.0f} us per block", flush=True)


if __name__ == "__main__":
    main()
