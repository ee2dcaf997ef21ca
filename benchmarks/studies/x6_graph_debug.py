#!/usr/bin/env python
"""Debug: te_linear_relprop_x6_f32 replayed from a HIP graph vs eager (same buffers), flags dumped after each replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.x6_bench import operands  # noqa: E402
from transformer_explainability_amd import _lib, ops  # noqa: E402

_lib.require_device()
lib = _lib.load()
dev = torch.device("cuda:0")
T, in_f, out_f = 64 * 197, 768, 2304
X, W, b, R, Y = operands(T, in_f, out_f, 1, dev)
R2 = R * 1.5 + 1e-4
planes = ops.x6_weight_planes(W, {})
al = lambda n: (n + 255) // 256 * 256      # noqa: E731
off_flags = al(lib.te_linear_x6_planes_bytes(T, in_f)) + al(lib.te_linear_x6_planes_bytes(T, out_f)) + (512 * 256 * 128 * 4)


def call(Rt, out, ws, flags=0):
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.te_linear_relprop_x6_f32(Rt.data_ptr(), None, 0, 1, X.data_ptr(), W.data_ptr(), planes.data_ptr(), None,
                                            Y.data_ptr(), b.data_ptr(), out.data_ptr(), T, in_f, out_f, flags, ws.data_ptr(),
                                            ws.numel(), st), "x6")


def flags_of(ws):
    z = ws[off_flags: off_flags + 4096].view(torch.int32).cpu()
    c = ws[off_flags + 65536: off_flags + 65536 + 4096].view(torch.int32).cpu()
    return int((z != 0).sum()), int((c != 0).sum())


ws = torch.zeros(lib.te_linear_relprop_x6_workspace_bytes(T, in_f, out_f), dtype=torch.uint8, device=dev)
out_e1, out_e2 = torch.empty_like(X), torch.empty_like(X)
call(R, out_e1, ws)
torch.cuda.synchronize()
print("eager 1 flags", flags_of(ws))
call(R2, out_e2, ws)
torch.cuda.synchronize()
print("eager 2 flags", flags_of(ws))

Rs = R.clone()
out_g = torch.empty_like(X)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    call(Rs, out_g, ws)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    call(Rs, out_g, ws)
    call(Rs, out_g, ws)      # two rules back to back inside one graph
for rep, Rt, ref in ((1, R, out_e1), (2, R2, out_e2), (3, R, out_e1), (4, R2, out_e2)):
    Rs.copy_(Rt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"replay {rep}: {dt * 1e3:.2f} ms, equal to eager: {bool(torch.equal(out_g, ref))}, max diff "
          f"{float((out_g - ref).abs().max()):.3e}, flags {flags_of(ws)}", flush=True)
# eager again on the same buffers
call(R, out_g, ws)
torch.cuda.synchronize()
print("eager after graph: equal", bool(torch.equal(out_g, out_e1)), flags_of(ws))
