#!/bin/bash
# round-3 GPU session 1: first light of the rewritten x6 Linear.relprop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/s1_build.log 2>&1
timeout 600 python benchmarks/x6_bench.py --config vit_b16 --iters 10 > gpurun_out/s1_x6_bench.log 2>&1
echo "x6_bench rc=$?" >> gpurun_out/s1_x6_bench.log
timeout 900 python -m pytest tests/test_gpu_rules.py -q -m gpu -k "linear" -x > gpurun_out/s1_pytest_linear.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1_pytest_linear.log
tail -30 gpurun_out/s1_x6_bench.log
tail -15 gpurun_out/s1_pytest_linear.log
