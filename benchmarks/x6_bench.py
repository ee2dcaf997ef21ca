#!/usr/bin/env python
"""Microbenchmark + on-device checks of the split-operand bf16 Linear.relprop (csrc/te_linear_x6.hip) on the Linear
shapes of the three single-GPU configurations.  Per shape and tile geometry: HIP-event time of the three phases
(|X| split, Z-pass, C-pass), executed bf16 TF, fp32-equivalent TF, against the fp32-MFMA kernels of te_linear.hip on the
same operands; checks: x6 vs fp64 (error no larger than the fp32-MFMA path's), the two tile geometries bit for bit,
a half batch bit for bit (rows independent of the stream-K cut), no expired hand-over wait.

    python benchmarks/x6_bench.py [--config vit_b16|vit_l16|bert_base|all] [--iters 20] [--check-rows 1024]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import transformer_explainability_amd as te  # noqa: E402
from transformer_explainability_amd import _lib, ops  # noqa: E402

CONFIGS = {
    # (T, [(name, in_f, out_f)])
    "vit_b16": (64 * 197, [("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]),
    "vit_l16": (32 * 577, [("qkv", 1024, 3072), ("proj", 1024, 1024), ("fc1", 1024, 4096), ("fc2", 4096, 1024)]),
    "bert_base": (32 * 512, [("qkv1", 768, 768), ("inter", 768, 3072), ("out", 3072, 768)]),
}


def operands(T, in_f, out_f, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = torch.randn(T, in_f, generator=g)
    W = 0.03 * torch.randn(out_f, in_f, generator=g)
    b = 0.1 * torch.randn(out_f, generator=g)
    R = 1e-3 * torch.randn(T, out_f, generator=g)
    X, W, b, R = (t.to(dev) for t in (X, W, b, R))
    Y = torch.nn.functional.linear(X, W, b)
    return X, W, b, R, Y


class Timer:
    def __init__(self):
        self.rows = {}

    def __call__(self, name, flops, nbytes):
        import contextlib

        @contextlib.contextmanager
        def cm():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            yield
            b.record()
            self.rows.setdefault(name, []).append((a, b, flops, nbytes))
        return cm()

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.rows.items():
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b, _, _ in evs)
            out[name] = dict(us_med=ts[len(ts) // 2], us_min=ts[0], flops=evs[0][2], bytes=evs[0][3], n=len(ts))
        return out


def bench_shape(T, in_f, out_f, iters, dev, tile):
    X, W, b, R, Y = operands(T, in_f, out_f, 1, dev)
    cache = {}
    ops.X6_TILE = tile
    res = {}
    for use_x6 in (True, False):
        ops.USE_LINEAR_X6 = use_x6
        for _ in range(3):
            ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
        torch.cuda.synchronize()
        t = Timer()
        ops.KERNEL_TIMER = t
        try:
            for _ in range(iters):
                ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
        finally:
            ops.KERNEL_TIMER = None
        res["x6" if use_x6 else "fp32"] = t.summary()
    ops.USE_LINEAR_X6 = True
    # whole rule, un-instrumented (one call = memset + split + Z + C back to back)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
    a.record()
    for _ in range(iters):
        ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
    e.record()
    torch.cuda.synchronize()
    res["x6_rule_us"] = a.elapsed_time(e) * 1e3 / iters
    return res


def check_shape(T, in_f, out_f, dev):
    """x6 vs fp64 and vs the fp32-MFMA path; tile geometries and a half batch bit for bit; hand-over flags clean."""
    from oracle import relprop_oracle as O
    X, W, b, R, Y = operands(T, in_f, out_f, 7, dev)
    X[1] = 0.0
    X[2] = X[2].abs() + 0.01
    W[:5] = -W[:5].abs() - 0.001                 # row 2 against these: every product negative -> cancellation fallback
    Y = torch.nn.functional.linear(X, W, b)
    cache = {}
    ops.USE_LINEAR_X6 = False
    fp32 = ops.linear_relprop(R, X, W, Y=Y, bias=b)
    ops.USE_LINEAR_X6 = True
    ops.X6_CHECK = True
    outs = {}
    for tile in (0, 1):
        ops.X6_TILE = tile
        outs[tile] = ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
    ops.X6_TILE = 0
    h = T // 2
    half = ops.linear_relprop(R[:h].contiguous(), X[:h].contiguous(), W, Y=Y[:h].contiguous(), bias=b, cache=cache)
    ops.X6_CHECK = False
    torch.cuda.synchronize()
    ref64 = O.linear_relprop(R.double().cpu(), X.double().cpu(), W.double().cpu(), 1.0, "ours")
    scale = float(ref64.abs().max())
    e6 = float((outs[0].cpu().double() - ref64).abs().max())
    e32 = float((fp32.cpu().double() - ref64).abs().max())
    return dict(T=T, in_f=in_f, out_f=out_f, finite=bool(torch.isfinite(outs[0]).all()),
                err_x6_vs_fp64=e6 / scale, err_fp32_vs_fp64=e32 / scale,
                tiles_bitwise=bool(torch.equal(outs[0], outs[1])), half_bitwise=bool(torch.equal(half, outs[0][:h])),
                max_abs_x6_vs_fp32=float((outs[0] - fp32).abs().max()) / scale)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_b16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check-rows", type=int, default=1100)
    ap.add_argument("--tiles", default="0,1")
    ap.add_argument("--skip-checks", action="store_true")
    a = ap.parse_args()
    _lib.require_device()
    dev = torch.device("cuda:0")
    names = list(CONFIGS) if a.config == "all" else a.config.split(",")
    if not a.skip_checks:
        for (T, i, o) in [(a.check_rows, 768, 2304), (a.check_rows, 3072, 768), (394, 768, 768), (300, 128, 384),
                          (70, 192, 128), (2 * 197, 1024, 4096)]:
            r = check_shape(T, i, o, dev)
            ok = (r["finite"] and r["tiles_bitwise"] and r["half_bitwise"]
                  and r["err_x6_vs_fp64"] <= 2.0 * r["err_fp32_vs_fp64"] + 1e-9)
            print(("CHECK ok   " if ok else "CHECK FAIL ") + json.dumps(r), flush=True)
    for name in names:
        T, shapes = CONFIGS[name]
        tot = {}
        for (lname, in_f, out_f) in shapes:
            for tile in [int(t) for t in a.tiles.split(",")]:
                if tile == 1 and not (out_f % 256 == 0 and in_f % 128 == 0):
                    continue
                r = bench_shape(T, in_f, out_f, a.iters, dev, tile)
                x6 = r["x6"]
                gemm = 2.0 * T * in_f * out_f
                line = dict(config=name, layer=lname, T=T, in_f=in_f, out_f=out_f, tile=("256" if tile == 0 else "128"),
                            split_us=x6["linear_x6_split"]["us_med"], z_us=x6["linear_x6_zpass"]["us_med"],
                            c_us=x6["linear_x6_cpass"]["us_med"], rule_us=r["x6_rule_us"],
                            z_bf16_tf=6 * gemm / x6["linear_x6_zpass"]["us_med"] * 1e-6,
                            c_bf16_tf=12 * gemm / x6["linear_x6_cpass"]["us_med"] * 1e-6,
                            fp32_z_us=r["fp32"].get("linear_zpass_fwd", {}).get("us_med"),
                            fp32_c_us=r["fp32"].get("linear_cpass", {}).get("us_med"))
                line["rule_fp32equiv_tf"] = 3 * gemm / line["rule_us"] * 1e-6
                line["rule_bf16_tf"] = 18 * gemm / line["rule_us"] * 1e-6
                print("BENCH " + json.dumps(line), flush=True)
                tot.setdefault(tile, []).append(line["rule_us"])
        for tile, v in tot.items():
            print(f"TOTAL {name} tile={'256' if tile == 0 else '128'}: one block's four rules {sum(v):.0f} us", flush=True)
    rc = te._lib.load()
    del rc


if __name__ == "__main__":
    main()
