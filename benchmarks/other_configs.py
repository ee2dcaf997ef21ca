#!/usr/bin/env python
"""Measurement infrastructure (not product code): throughput of the other BASELINE.json configurations on one MI355X --
configs[2] ViT-L/16 384^2 batch 32 and configs[3] BERT-base 512 tokens batch 32 -- with the same step definition as
bench.py (inputs resident, W warm-up steps, K timed steps between synchronises).  One JSON line per configuration.

    python benchmarks/other_configs.py [--steps 5] [--warmup 2] [--prune]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transformer_explainability_amd as te  # noqa: E402
from transformer_explainability_amd import bert, vit  # noqa: E402
from transformer_explainability_amd.generators import LRP, Generator  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prune", action="store_true")
    args = ap.parse_args()
    d = torch.device("cuda:0")
    tuned = te.enable_tuned_gemms()
    torch.manual_seed(0)
    B = 32

    model = vit.vit_large_patch16_224(img_size=384).eval().to(d)
    x = torch.randn(B, 3, 384, 384, device=d)
    lrp = LRP(model, prune=args.prune)
    s = timed(lambda: lrp.generate_LRP(x, start_layer=1), args.steps, args.warmup)
    print(json.dumps({"workload": "ViT-L/16 384^2 batch 32 (BASELINE.json configs[2])", "maps_per_s": B / s,
                      "ms_per_step": s * 1e3, "tuned_stock_gemms": tuned, "prune": args.prune}), flush=True)
    del model, lrp, x
    torch.cuda.empty_cache()

    model = bert.BertForSequenceClassification(bert.BertConfigLite(num_labels=2)).eval().to(d)
    ids = torch.randint(1000, 20000, (B, 512), device=d)
    mask = torch.ones(B, 512, device=d)
    for sl in (0, 11):
        gen = Generator(model, prune=args.prune)
        s = timed(lambda: gen.generate_LRP(ids, mask, start_layer=sl), args.steps, args.warmup)
        print(json.dumps({"workload": f"BERT-base 512 tokens batch 32, start_layer={sl} (BASELINE.json configs[3])",
                          "maps_per_s": B / s, "ms_per_step": s * 1e3, "tuned_stock_gemms": tuned,
                          "prune": args.prune}), flush=True)


if __name__ == "__main__":
    main()
