#!/usr/bin/env python
"""GPU utility: extend the committed TunableOp selection (transformer-explainability_amd/tuning/tunableop_gfx950.csv) with
the stock-GEMM shapes of the other BASELINE.json configurations.  Writes gpurun_out/tunableop_gfx950.csv (start = the
committed file, new shapes appended by PyTorch as they are tuned); copy it over the committed one to adopt it.

    python scripts/tune_gemms.py [vit_b16] [vit_l16_384] [bert_base_512]
"""
import os
import shutil
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transformer_explainability_amd as te  # noqa: E402
from transformer_explainability_amd import bert, vit  # noqa: E402
from transformer_explainability_amd.generators import LRP, Generator  # noqa: E402


def main():
    which = sys.argv[1:] or ["vit_b16", "vit_l16_384", "bert_base_512"]
    out = os.path.join(ROOT, "gpurun_out", "tunableop_gfx950.csv")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    shutil.copyfile(os.path.join(ROOT, "transformer-explainability_amd", "tuning", "tunableop_gfx950.csv"), out)
    assert te.enable_tuned_gemms(out, tune=True)
    import torch.cuda.tunable as tunable
    tunable.set_max_tuning_duration(15)
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    for name in which:
        t0 = time.time()
        if name == "vit_b16":
            m = vit.vit_base_patch16_224().eval().to(d)
            LRP(m).generate_LRP(torch.randn(64, 3, 224, 224, device=d), start_layer=1)
        elif name == "vit_l16_384":
            m = vit.vit_large_patch16_224(img_size=384).eval().to(d)
            LRP(m).generate_LRP(torch.randn(32, 3, 384, 384, device=d), start_layer=1)
        elif name == "bert_base_512":
            m = bert.BertForSequenceClassification(bert.BertConfigLite(num_labels=2)).eval().to(d)
            ids = torch.randint(1000, 20000, (32, 512), device=d)
            Generator(m).generate_LRP(ids, torch.ones(32, 512, device=d), start_layer=0)
        else:
            raise SystemExit(f"unknown workload {name}")
        torch.cuda.synchronize()
        del m
        torch.cuda.empty_cache()
        print(f"{name}: tuned in {time.time() - t0:.1f} s", flush=True)
    print(open(out).read())


if __name__ == "__main__":
    main()
