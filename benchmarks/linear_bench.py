#!/usr/bin/env python
"""Measurement infrastructure (not product code): per-shape rate of the two Linear.relprop kernels on one MI355X,
next to the register-only fp32-MFMA rate the same chip sustains (benchmarks/mfma_peak.hip).

    [TE_LINEAR_TILE=128x128|128x64|64x64] python benchmarks/linear_bench.py [--batch 64] [--clock] [--lib ...]   (GPU box)
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
    ap.add_argument("--clock", action="store_true", help="sample the shader clock while each kernel runs")
    ap.add_argument("--lib", default=None, help="alternative build of the C ABI (benchmarks/libte_ablate*.so)")
    args = ap.parse_args()
    import torch
    from transformer_explainability_amd import _lib
    lib = _lib.load()
    _lib.require_device()
    if args.lib:
        lib = ctypes.CDLL(os.path.abspath(args.lib))
        for n in ("te_linear_zpass_f32", "te_linear_cpass_f32"):
            getattr(lib, n).restype = ctypes.c_int
            getattr(lib, n).argtypes = _lib.SIGNATURES[n][1]
    d = torch.device("cuda:0")
    res = {"env_TE_LINEAR_TILE": os.environ.get("TE_LINEAR_TILE", "auto"), "lib": args.lib or "product"}

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

    pk = ctypes.CDLL(os.path.join(ROOT, "benchmarks", "libmfma_peak.so"))
    pk.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    pk.clock_probe_launch.restype = None
    side = torch.cuda.Stream()
    probe_out = torch.zeros(2, dtype=torch.int64, device=d)

    def shader_mhz(work, spin_us=1500):
        """run `work()` (enqueues ~3 ms of kernels on the current stream) with the clock probe on a side stream"""
        torch.cuda.synchronize()
        work()                                       # get going first
        pk.clock_probe_launch(probe_out.data_ptr(), spin_us, side.cuda_stream)
        work()
        torch.cuda.synchronize()
        c, r = [int(v) for v in probe_out.cpu()]
        return c / max(r, 1) * 100.0

    if args.clock:
        res["shader_mhz_idle"] = shader_mhz(lambda: None)
        print(f"shader clock, idle chip: {res['shader_mhz_idle']:.0f} MHz", flush=True)

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
        bias = torch.randn(out_f, device=d) * 0.1
        Y = torch.nn.functional.linear(X, W, bias)
        legs = [("zpass", lib.te_linear_zpass_f32, (R, X, W, S), 2.0 * T * (2 * in_f) * out_f),
                ("cpass", lib.te_linear_cpass_f32, (S, X, W, out), 2.0 * T * (2 * in_f) * out_f)]
        if not args.lib or "prev" in args.lib:
            lib.te_linear_zpass_fwd_f32.restype = ctypes.c_int
            lib.te_linear_zpass_fwd_f32.argtypes = _lib.SIGNATURES["te_linear_zpass_fwd_f32"][1]
            legs.insert(1, ("zfwd", lib.te_linear_zpass_fwd_f32, (R, X, W, Y, bias, S), 2.0 * T * in_f * out_f))
        for kname, fn, a, flops in legs:
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
            if args.clock:
                n = max(2, int(2500 / us))
                mhz = shader_mhz(lambda: [fn(*ptrs, T, in_f, out_f, st) for _ in range(n)])
                res[f"{name}.{kname}"]["shader_mhz"] = mhz
                res[f"{name}.{kname}"]["frac_of_clock_peak"] = tf / (157.3 * mhz / 2400.0)
                print(f"      shader clock under this kernel: {mhz:.0f} MHz -> {tf / (157.3 * mhz / 2400.0) * 100:.1f} % of "
                      f"the MFMA rate at that clock", flush=True)
            if kname != "zpass" or args.lib:      # the shipped pair is zfwd + cpass
                tot_f += flops
                tot_t += us * 1e-6
            print(f"{name:5s} {kname}  T={T} in={in_f} out={out_f}: {us:8.1f} us  {tf:6.1f} TF", flush=True)
    res["block_total"] = {"us": tot_t * 1e6, "tflops": tot_f / tot_t / 1e12}
    print(f"one block's 4 Linear rules (Z-pass from the forward output + C-pass): {tot_t * 1e6:.0f} us, "
          f"{tot_f / tot_t / 1e12:.1f} TF on {tot_f / 1e9:.0f} algorithmic GFLOP")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
