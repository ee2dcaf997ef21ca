#!/bin/bash
# trip 2: x6 tests incl. the 128 x 128 geometry, geometry variants per shape (ViT-B, BERT, ViT-L)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rules.py -q -m gpu -k "x6" > gpurun_out/t2_tests.log 2>&1
grep -v amdgpu gpurun_out/t2_tests.log | tail -15
timeout 400 python benchmarks/x6_variants.py --iters 10 --variants base,g128,g64,g256 > gpurun_out/t2_variants_vitb.log 2>&1
grep -v amdgpu.ids gpurun_out/t2_variants_vitb.log | tail -24
timeout 400 python benchmarks/x6_variants.py --iters 6 --variants base,g128,g64 --config bert_base > gpurun_out/t2_variants_bert.log 2>&1
grep -v amdgpu.ids gpurun_out/t2_variants_bert.log | tail -14
timeout 400 python benchmarks/x6_variants.py --iters 4 --variants base,g128,g64 --config vit_l16 > gpurun_out/t2_variants_vitl.log 2>&1
grep -v amdgpu.ids gpurun_out/t2_variants_vitl.log | tail -18
