#!/usr/bin/env python
"""Measurement infrastructure (not product code): per-shape rate of the two Linear.relprop kernels on one MI355X,
next to the register-only fp32-MFMA rate the same chip sustains (benchmarks/mfma_peak.hip).

    python benchmarks/linear_bench.py [--batch 64] [--bn 0|64|128]   (run on the GPU box)
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tokens", type=int, default=197)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-peak", action="store_true")
    args = ap.parse_args()
    import torch
    from transformer_explainability_amd import _lib
    lib = _lib.load()
    _lib.require_device()
    d = torch.device("cuda:0")
    res = {"env_TE_LINEAR_BN": os.environ.get("TE_LINEAR_BN", "auto")}

    if not args.skip_peak:
        pk = ctypes.CDLL(os.path.join(ROOT, "benchmarks", "libmfma_peak.so"))
        pk.mfma_peak_tflops.restype = ctypes.c_double
        pk.mfma_peak_tflops.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]
        scratch = torch.empty(256 * 8 * 256, device=d)
        seed = torch.rand(64, device=d) - 0.5
        ms = ctypes.c_double()
        for nacc, bpc in ((4, 1), (4, 2), (8, 1)):
            tf = pk.mfma_peak_tflops(nacc, bpc, 4000, scratch.data_ptr(), seed.data_ptr(), ctypes.byref(ms))
            res[f"mfma_peak_nacc{nacc}_blocks{bpc}"] = {"tflops": tf, "ms": ms.value}
            print(f"register-only mfma_f32_32x32x2: {nacc} accumulators, {bpc} block(s)/CU: {tf:7.1f} TF ({ms.value:.2f} ms)",
                  flush=True)

    T = args.batch * args.tokens
    shapes = [("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]
    st = torch.cuda.current_stream().cuda_stream
    tot_f = tot_t = 0.0
    for name, in_f, out_f in shapes:
        X = torch.randn(T, in_f, device=d)
        W = torch.randn(out_f, in_f, device=d) * 0.02
        R = torch.randn(T, out_f, device=d) * 0.01
        S = torch.empty(T, out_f, device=d)
        out = torch.empty(T, in_f, device=d)
        flops = 2.0 * T * (2 * in_f) * out_f
        for kname, fn, a in (("zpass", lib.te_linear_zpass_f32, (R, X, W, S)), ("cpass", lib.te_linear_cpass_f32, (S, X, W, out))):
            ptrs = [t.data_ptr() for t in a]
            for _ in range(2):
                _lib.check(fn(*ptrs, T, in_f, out_f, st), kname)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn(*ptrs, T, in_f, out_f, st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.reps * 1e3
            tf = flops / (us * 1e-6) / 1e12
            res[f"{name}.{kname}"] = {"us": us, "tflops": tf}
            tot_f += flops
            tot_t += us * 1e-6
            print(f"{name:5s} {kname}  T={T} in={in_f} out={out_f}: {us:8.1f} us  {tf:6.1f} TF", flush=True)
    res["block_total"] = {"us": tot_t * 1e6, "tflops": tot_f / tot_t / 1e12}
    print(f"one block's 4 Linear rules: {tot_t * 1e6:.0f} us, {tot_f / tot_t / 1e12:.1f} TF")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
