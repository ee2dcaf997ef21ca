#!/usr/bin/env python
"""GPU measurement: what a write-only stream reaches on this part (fill_ of 0.8 / 1.4 GB) next to a copy -- the yardstick for the
attention forward producers, whose traffic is 94 % stores (two N x N tensors out, q / k / v in)."""
import torch


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


d = torch.device("cuda:0")
for mb in (805, 1364):
    a = torch.empty(mb * 250000, dtype=torch.float32, device=d)
    b = torch.empty_like(a)
    us = t(lambda: a.fill_(1.5))
    print(f"fill_ {mb} MB: {us:7.1f} us  {mb / us * 1e-6 * 1e6 / 1e6:.2f} TB/s written")
    us = t(lambda: b.copy_(a))
    print(f"copy_ {mb} MB: {us:7.1f} us  {2 * mb / us:.2f} TB/s moved (read + write)".replace("TB/s", "MB/us = TB/s"))
    us = t(lambda: torch.add(a, 1.0, out=b))
    print(f"add   {mb} MB: {us:7.1f} us  {2 * mb / us:.2f} MB/us = TB/s moved")
