#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python benchmarks/x6_bench.py --config vit_b16 --iters 10 --tiles 0 > gpurun_out/s16_x6_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/s16_x6_bench.log | cut -c1-330 | grep "CHECK\|TOTAL\|rc=\|rule_us" | sed 's/"fp32_z_us.*rule_fp32/ rule_fp32/'
( timeout 300 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "linear or x6" 2>&1 | tail -5 ) 
( timeout 200 python bench.py --cpu-baseline off --steps 12 > gpurun_out/s16_bench.json 2> gpurun_out/s16_bench.err ); cut -c1-230 gpurun_out/s16_bench.json; tail -2 gpurun_out/s16_bench.err
