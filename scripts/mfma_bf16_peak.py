#!/usr/bin/env python
"""GPU measurement: register-only MFMA rate of this MI355X, fp32 (v_mfma_f32_32x32x2_f32) next to bf16
(v_mfma_f32_32x32x16_bf16) -- the ceiling a split-operand fp32 GEMM (six bf16 MFMAs per fp32 product)
would be priced against.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC benchmarks/mfma_peak.hip -o benchmarks/libmfma_peak.so"""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pk = ctypes.CDLL(os.path.join(ROOT, "benchmarks", "libmfma_peak.so"))
args = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
for f in (pk.mfma_peak_tflops, pk.mfma_peak_bf16_tflops):
    f.restype, f.argtypes = ctypes.c_double, args
d = torch.device("cuda:0")
scratch = torch.empty(256 * 8 * 256, device=d)
seed = torch.rand(64, device=d) - 0.5
ms = ctypes.c_double()
for nacc, bpc in ((4, 1), (4, 2), (8, 1), (8, 2)):
    f32 = pk.mfma_peak_tflops(nacc, bpc, 4000, scratch.data_ptr(), seed.data_ptr(), ctypes.byref(ms))
    t32 = ms.value
    b16 = pk.mfma_peak_bf16_tflops(nacc, bpc, 4000, scratch.data_ptr(), seed.data_ptr(), ctypes.byref(ms))
    print(f"{nacc} accumulators, {bpc} block(s)/CU: fp32 32x32x2 {f32:7.1f} TF ({t32:.2f} ms) | bf16 32x32x16 {b16:7.1f} TF "
          f"({ms.value:.2f} ms) | bf16 / 6 = {b16 / 6:6.1f} TF fp32-equivalent = {b16 / 6 / f32:.2f} x fp32", flush=True)
