#!/usr/bin/env python
"""GPU measurement (study, benchmarks/studies/gemm_bf16x6.hip): accuracy and rate of an fp32-accurate GEMM made of six
bf16 MFMAs per fp32 product, next to torch's fp32 GEMM, on the Linear-rule shapes of ViT-B/16 at batch 64.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC benchmarks/studies/gemm_bf16x6.hip -o benchmarks/studies/libgemm_bf16x6.so
    python scripts/gemm_bf16x6_bench.py
"""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "benchmarks", "studies", "libgemm_bf16x6.so"))
P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.split3_f32.restype, lib.split3_f32.argtypes = I, [P, P, I64, I64, P]
lib.gemm_x6_f32.restype, lib.gemm_x6_f32.argtypes = I, [P, P, P, I, I, I, P]
d = torch.device("cuda:0")


def stream():
    return torch.cuda.current_stream(d).cuda_stream


def split(x):
    R, K = x.shape
    out = torch.empty((R, K // 32, 3, 32), dtype=torch.bfloat16, device=d)
    assert lib.split3_f32(x.data_ptr(), out.data_ptr(), R, K, stream()) == 0
    return out


def gemm(a_s, b_s, M, N, K):
    c = torch.empty((M, N), dtype=torch.float32, device=d)
    assert lib.gemm_x6_f32(a_s.data_ptr(), b_s.data_ptr(), c.data_ptr(), M, N, K, stream()) == 0
    return c


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    torch.manual_seed(0)
    # correctness on a small problem: the split is exact, the product fp32-class
    A = torch.randn(256, 128, device=d) * torch.rand(256, 1, device=d) * 4
    B = torch.randn(384, 128, device=d)
    a_s, b_s = split(A), split(B)
    back = a_s.float().sum(2).reshape(256, 128)
    print("split exact:", bool(torch.equal(back, A)), flush=True)
    exact = A.double() @ B.double().t()
    c6, c32 = gemm(a_s, b_s, 256, 384, 128), A @ B.t()
    den = float(exact.abs().max())
    print(f"small GEMM  max|err|/max|exact|: x6 {float((c6.double() - exact).abs().max()) / den:.2e}   torch fp32 "
          f"{float((c32.double() - exact).abs().max()) / den:.2e}", flush=True)
    # the Linear-rule shapes (T = 64 x 197 rows padded to whole 128-row tiles)
    T = 12672
    for name, N, K in (("qkv Z-pass", 2304, 768), ("proj", 768, 768), ("fc1 Z-pass", 3072, 768), ("fc2 Z-pass", 768, 3072),
                       ("fc1 C-pass (K = out)", 768, 3072)):
        A = torch.randn(T, K, device=d).abs_()
        B = torch.randn(N, K, device=d).abs_() * 0.05
        a_s, b_s = split(A), split(B)
        c6 = gemm(a_s, b_s, T, N, K)
        c32 = A @ B.t()
        rows = slice(0, 512)
        exact = A[rows].double() @ B.double().t()
        den = float(exact.abs().max())
        e6, e32 = float((c6[rows].double() - exact).abs().max()) / den, float((c32[rows].double() - exact).abs().max()) / den
        t6 = timeit(lambda: gemm(a_s, b_s, T, N, K))
        t32 = timeit(lambda: A @ B.t())
        ts = timeit(lambda: split(A))
        fl = 2.0 * T * N * K
        print(f"{name:22s} [{T} x {N} x {K}]  x6 {t6 * 1e6:7.1f} us = {fl / t6 / 1e12:6.1f} TF fp32-equivalent (err {e6:.1e}) | "
              f"torch fp32 {t32 * 1e6:7.1f} us = {fl / t32 / 1e12:6.1f} TF (err {e32:.1e}) | split of A {ts * 1e6:6.1f} us",
              flush=True)


if __name__ == "__main__":
    main()
