#!/bin/bash
# trip 4: per-shape probe of the step under the geometry policy vs pins; x6 tests after the K-segment split
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -x -q -m gpu -k "x6 or linear or bert_layout" > gpurun_out/t4_tests.log 2>&1
grep -v amdgpu gpurun_out/t4_tests.log | tail -8
timeout 300 python benchmarks/step_probe.py --tag policy > gpurun_out/t4_probe_policy.log 2>&1
grep -E "^(STEP|GRP|TOT)" gpurun_out/t4_probe_policy.log | grep -E "STEP|TOT|linear"
TE_X6_FLAGS=0x400 timeout 300 python benchmarks/step_probe.py --tag z128 > gpurun_out/t4_probe_z128.log 2>&1
grep -E "^(STEP|GRP|TOT)" gpurun_out/t4_probe_z128.log | grep -E "STEP|TOT|linear_x6_zpass"
timeout 300 python benchmarks/x6_variants.py --iters 10 --variants base,g128,g64,g256 > gpurun_out/t4_variants_vitb.log 2>&1
grep -v amdgpu.ids gpurun_out/t4_variants_vitb.log | tail -22
