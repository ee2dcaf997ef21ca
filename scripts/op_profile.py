#!/usr/bin/env python
"""GPU diagnostic: which torch (ATen) ops around the HIP relprop kernels cost device time in one generate_LRP step
(ViT-B/16, batch 64)?  Prints the ops by self device time with their input shapes."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer_explainability_amd import vit  # noqa: E402
from transformer_explainability_amd.generators import LRP  # noqa: E402


def main():
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    model = vit.vit_base_patch16_224().eval().to(d)
    x = torch.randn(64, 3, 224, 224, device=d)
    lrp = LRP(model)
    for _ in range(2):
        lrp.generate_LRP(x, start_layer=1)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        lrp.generate_LRP(x, start_layer=1)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=60,
                                                             max_name_column_width=40, max_shapes_column_width=70))


if __name__ == "__main__":
    main()
