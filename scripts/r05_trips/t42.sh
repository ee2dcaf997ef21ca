#!/bin/bash
# the driver's own invocation: python bench.py (N = 1, default steps / warm-up)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 200 python bench.py > gpurun_out/bench_b64_default.json 2> gpurun_out/bench_b64_default.err )
echo "rc=$?"; cut -c1-330 gpurun_out/bench_b64_default.json; tail -2 gpurun_out/bench_b64_default.err
