#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python benchmarks/x6_graph_debug2.py > gpurun_out/s9_graph_debug2.log 2>&1
grep -v amdgpu.ids gpurun_out/s9_graph_debug2.log | tail -40
