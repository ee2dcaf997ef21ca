#!/usr/bin/env python
"""Microbenchmark of the plain x6 GEMM (te_gemm_x6_f32: y = x W^T + b and d_x = d_y W on split-operand bf16 MFMAs) on the
Linear shapes of ViT-B/16 batch 64 against torch's fp32 GEMM on the same operands: operand split and product timed
separately (HIP events around 20 back-to-back launches each).

    python benchmarks/x6_gemm_bench.py [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from transformer_explainability_amd import _lib, ops  # noqa: E402

T = 64 * 197
SHAPES = [("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    _lib.require_device()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    tot = dict(x6=0.0, x6_split=0.0, torch=0.0)
    for name, in_f, out_f in SHAPES:
        W = (0.03 * torch.randn(out_f, in_f, generator=g)).to(dev)
        bias = (0.1 * torch.randn(out_f, generator=g)).to(dev)
        for direction in ("forward", "backward"):
            K, M = (in_f, out_f) if direction == "forward" else (out_f, in_f)
            X = torch.randn(T, K, generator=g).to(dev)
            wp = ops.x6_matrix_planes(W, direction == "backward")
            xp = torch.empty(lib.te_linear_x6_planes_bytes(T, K), dtype=torch.uint8, device=dev)
            ws = torch.empty(lib.te_gemm_x6_workspace_bytes(T, K, M), dtype=torch.uint8, device=dev)
            out = torch.empty(T, M, device=dev)
            s = torch.cuda.current_stream().cuda_stream
            b_ptr = bias.data_ptr() if direction == "forward" else None

            def split():
                _lib.check(lib.te_linear_x6_split_matrix_f32(X.data_ptr(), T, K, 0, xp.data_ptr(), xp.numel(), s), "split")

            def gemm():
                _lib.check(lib.te_gemm_x6_f32(X.data_ptr(), xp.data_ptr(), wp.data_ptr(), b_ptr, out.data_ptr(), T, K, M,
                                              ops.X6_TILE | ops.X6_FLAGS, ops.x6_status(dev).data_ptr(), ws.data_ptr(), ws.numel(),
                                              s), "gemm")

            Wt = W if direction == "forward" else W.t().contiguous()

            def stock():
                return torch.nn.functional.linear(X, Wt, bias if direction == "forward" else None)

            split()
            t_split, t_gemm, t_stock = timed(split, a.iters), timed(gemm, a.iters), timed(stock, a.iters)
            ref = stock().double()
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            flops = 2.0 * T * K * M
            print("GEMM " + json.dumps(dict(layer=name, direction=direction, T=T, K=K, M=M, split_us=round(t_split, 1),
                                            x6_us=round(t_gemm, 1), torch_us=round(t_stock, 1),
                                            x6_fp32equiv_tf=round(flops / t_gemm * 1e-6, 1),
                                            x6_bf16_tf=round(6 * flops / t_gemm * 1e-6, 1),
                                            torch_tf=round(flops / t_stock * 1e-6, 1), rel_diff=err)), flush=True)
            if os.environ.get("TE_X6_G_PROF") == "1":      # study build: in-kernel wall-clock stamps of the last launch
                al = lambda n: (n + 255) // 256 * 256      # noqa: E731
                off = al(lib.te_linear_x6_planes_bytes(T, K)) + 512 * 256 * 128 * 4 + 8192
                raw = ws[off: off + 512 * 64].view(torch.int64).view(512, 8).cpu()
                act = raw[raw[:, 4] > 0].double()
                t0 = act[:, 7].min()
                m = act.mean(0) * 0.01
                print("PROF " + json.dumps(dict(layer=name, direction=direction, wgs=int(act.shape[0]),
                                                span_us=round(float(((act[:, 7] - t0) + act[:, 6]).max() * 0.01), 1),
                                                loop_us=round(float(m[0]), 1), epi_us=round(float(m[1]), 1),
                                                pub_us=round(float(m[2]), 1), wait_us=round(float(m[3]), 1),
                                                steps=round(float(act[:, 4].mean()), 1), nepi=round(float(act[:, 5].mean()), 2),
                                                ns_per_step=round(float(act[:, 0].sum() / act[:, 4].sum() * 10), 1),
                                                wg_total_us=round(float(m[6]), 1), start_skew_us=round(float((act[:, 7] - t0).max() * 0.01), 1))), flush=True)
            tot["x6"] += t_gemm
            tot["x6_split"] += t_split
            tot["torch"] += t_stock
    print("TOTAL one block, forward + backward: x6 products %.0f us + splits %.0f us; torch %.0f us" %
          (tot["x6"], tot["x6_split"], tot["torch"]))


if __name__ == "__main__":
    main()
