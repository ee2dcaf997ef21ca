#!/usr/bin/env python
"""Main-loop ablations of x6_kernel (library built with TE_BUILD_DEFINES=TE_X6_STUDY): HIP-event time of the Z- and
C-pass of the ViT-B Linear shapes for study 0 (shipped) .. 4 (see csrc/te_linear_x6.hip).  Timing only: the ablations'
results are garbage.   python benchmarks/x6_study.py [--iters 10] [--studies 0,1,2,3,4]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.x6_bench import CONFIGS, Timer, operands  # noqa: E402
from transformer_explainability_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--studies", default="0,1,2,3,4")
ap.add_argument("--config", default="vit_b16")
ap.add_argument("--once", action="store_true", help="one un-instrumented call per shape (for rocprofv3 --pmc runs)")
a = ap.parse_args()
_lib.require_device()
dev = torch.device("cuda:0")
T, shapes = CONFIGS[a.config]
NAMES = {0: "shipped", 1: "no-global-loads", 2: "no-loads-no-barrier", 3: "mfma-only", 4: "no-epilogue"}
for (lname, in_f, out_f) in shapes:
    X, W, b, R, Y = operands(T, in_f, out_f, 1, dev)
    cache = {}
    if a.once:
        for _ in range(3):
            ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
        torch.cuda.synchronize()
        continue
    gemm = 2.0 * T * in_f * out_f
    for st in [int(s) for s in a.studies.split(",")]:
        ops.X6_TILE = st << 5
        for _ in range(2):
            ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
        torch.cuda.synchronize()
        t = Timer()
        ops.KERNEL_TIMER = t
        for _ in range(a.iters):
            ops.linear_relprop(R, X, W, Y=Y, bias=b, cache=cache)
        ops.KERNEL_TIMER = None
        s = t.summary()
        z, c = s["linear_x6_zpass"]["us_med"], s["linear_x6_cpass"]["us_med"]
        print("STUDY " + json.dumps(dict(layer=lname, in_f=in_f, out_f=out_f, study=st, name=NAMES[st], z_us=round(z, 1),
                                         c_us=round(c, 1), z_tf=round(6 * gemm / z * 1e-6), c_tf=round(12 * gemm / c * 1e-6))),
              flush=True)
    ops.X6_TILE = 0
