#!/bin/bash
# trip 7: why is the step slow? per-shape probe inside bench.py vs the eager step probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python benchmarks/step_probe.py --tag eager > gpurun_out/t7_probe.log 2>&1
grep -E "^(STEP|GRP|TOT)" gpurun_out/t7_probe.log | grep -E "STEP|TOT|linear" | cut -c1-200
TE_BENCH_DUMP=1 timeout 400 python bench.py --steps 6 --cpu-baseline off > gpurun_out/t7_bench.json 2> gpurun_out/t7_bench.err
grep -E "probe linear|timed|comparison" gpurun_out/t7_bench.err
TE_BENCH_DUMP=1 timeout 400 python bench.py --steps 6 --cpu-baseline off --graph off --overlap-backward off > gpurun_out/t7_bench_eager.json 2> gpurun_out/t7_bench_eager.err
grep -E "probe linear|timed|comparison" gpurun_out/t7_bench_eager.err
