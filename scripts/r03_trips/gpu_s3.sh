#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python benchmarks/x6_bench.py --config vit_b16 --iters 10 > gpurun_out/s3_x6_bench.log 2>&1
echo "x6_bench rc=$?" >> gpurun_out/s3_x6_bench.log
grep -v amdgpu.ids gpurun_out/s3_x6_bench.log | cut -c1-600
