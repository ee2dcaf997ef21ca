#!/bin/bash
# configs[4] in full on one GPU: 50 000 images = 195 batches of 256 + the short last batch of 80 (run eagerly), one gather
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 320 python bench.py --config sweep50k --cpu-baseline off --no-roofline > gpurun_out/bench_sweep50k_full.json 2> gpurun_out/bench_sweep50k_full.err )
echo "rc=$?"; cut -c1-500 gpurun_out/bench_sweep50k_full.json; tail -6 gpurun_out/bench_sweep50k_full.err
