#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 200 python benchmarks/x6_gemm_bench.py 2>&1 | tail -12 ) > gpurun_out/s19_x6_gemm_bench.log
cut -c1-330 gpurun_out/s19_x6_gemm_bench.log
