#!/usr/bin/env python
"""GPU diagnosis: where do the device-to-device memcpys (__amd_rocclr_copyBuffer in the rocprofv3 tables) of a step come from?
One eager step of a bench configuration under torch.profiler with Python stacks; prints the call sites of aten::copy_ / aten::clone /
aten::contiguous with the bytes they move.    python scripts/find_copies.py bert_base_512 4 [fp32]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

cfg, B = sys.argv[1], int(sys.argv[2])
args = bench.parse_args(["--config", cfg, "--batch", str(B), "--cpu-baseline", "off", "--parity", "off", "--no-roofline"])
args.overlap_backward = "on" if args.overlap_backward == "auto" else args.overlap_backward
dev = torch.device("cuda:0")
from transformer_explainability_amd import ops  # noqa: E402
ops.USE_FUSED_PRODUCERS = True      # (bench.main: --producers fused, the default)
if len(sys.argv) > 3 and sys.argv[3] == "fp32":      # the comparison run of the bench line: Linear layers on the fp32-MFMA kernels
    ops.USE_LINEAR_X6 = False
wl = bench.Workload(args, 0, dev)
for _ in range(2):
    wl.eager(*wl.inputs)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    wl.eager(*wl.inputs)
    torch.cuda.synchronize()
sites = collections.Counter()
byt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if ("transformer" in s or "bench.py" in s) and "profiler" not in s][:4]
        shape = ev.input_shapes[0] if ev.input_shapes else None
        key = (ev.name, str(shape), " <- ".join(s.split("/")[-1] for s in st))
        sites[key] += 1
        n = 1
        for d in (shape or []):
            n *= d
        byt[key] += 4 * n
print("memcpy-like kernels:", sum(1 for ev in prof.events() if "Memcpy" in ev.name or "copyBuffer" in ev.name))
for key, c in sorted(sites.items(), key=lambda kv: -byt[kv[0]])[:25]:
    print(f"{c:4d} x {key[0]:16s} {key[1]:28s} {byt[key] / 1e6:9.1f} MB  {key[2]}")
