#!/usr/bin/env python
"""Where a workgroup of x6_kernel spends its time (library built with TE_BUILD_DEFINES=TE_X6_STUDY): study 5 = the
shipped kernel with wall-clock stamps (s_memrealtime, 10 ns ticks) around its phases, study 6 = the same without the
epilogue.  Per pass: mean / max over workgroups of main-loop time, epilogue time, publish time, wait time, steps,
main-loop ns per K-step, kernel span.   python benchmarks/x6_prof.py [--tile 0|1]"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.x6_bench import CONFIGS, operands  # noqa: E402
from transformer_explainability_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--config", default="vit_b16")
ap.add_argument("--studies", default="5,6")
a = ap.parse_args()
_lib.require_device()
lib = _lib.load()
dev = torch.device("cuda:0")
T, shapes = CONFIGS[a.config]
al = lambda n: (n + 255) // 256 * 256      # noqa: E731
for (lname, in_f, out_f) in shapes:
    X, W, b, R, Y = operands(T, in_f, out_f, 1, dev)
    planes = ops.x6_weight_planes(W, {})
    out = torch.empty_like(X)
    ws = torch.zeros(lib.te_linear_relprop_x6_workspace_bytes(T, in_f, out_f), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    off_flags = al(lib.te_linear_x6_planes_bytes(T, in_f)) + al(lib.te_linear_x6_planes_bytes(T, out_f)) + (512 * 256 * 128 * 4)
    for study in [int(s) for s in a.studies.split(",")]:
        if a.tile == 1 and study != 5:
            continue
        for rep in range(2):
            for phase in (4, 8, 16):
                _lib.check(lib.te_linear_relprop_x6_f32(R.data_ptr(), None, 0, 1, X.data_ptr(), W.data_ptr(), planes.data_ptr(),
                                                        None, Y.data_ptr(), b.data_ptr(), out.data_ptr(), T, in_f, out_f,
                                                        a.tile | phase | (study << 5), ops.x6_status(dev).data_ptr(), ws.data_ptr(),
                                                        ws.numel(), st), "x6")
        torch.cuda.synchronize()
        for pi, pname in enumerate(("zpass", "cpass")):
            raw = ws[off_flags + pi * 65536 + 8192: off_flags + pi * 65536 + 8192 + 512 * 64].view(torch.int64).view(512, 8).cpu()
            act = raw[raw[:, 4] > 0].double()
            if act.numel() == 0:
                continue
            t0 = act[:, 7].min()
            span = ((act[:, 7] - t0) + act[:, 6]).max() * 0.01
            m = act.mean(0) * 0.01          # us
            mx = act.max(0).values * 0.01
            print("PROF " + json.dumps(dict(layer=lname, pass_=pname, study=study, tile=a.tile, wgs=int(act.shape[0]),
                                            span_us=round(float(span), 1), loop_us=[round(float(m[0]), 1), round(float(mx[0]), 1)],
                                            epi_us=[round(float(m[1]), 1), round(float(mx[1]), 1)],
                                            pub_us=[round(float(m[2]), 1), round(float(mx[2]), 1)],
                                            wait_us=[round(float(m[3]), 1), round(float(mx[3]), 1)],
                                            steps=round(float(act[:, 4].mean()), 1), nepi=[round(float(act[:, 5].mean()), 2), float(act[:, 5].max())],
                                            ns_per_step=round(float(act[:, 0].sum() / act[:, 4].sum() * 10), 1),
                                            wg_total_us=[round(float(m[6]), 1), round(float(mx[6]), 1)])), flush=True)
