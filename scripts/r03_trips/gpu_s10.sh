#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 200 python benchmarks/x6_graph_debug3.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s10_debug3.log
( DBG_FUSED=0 timeout 200 python benchmarks/x6_graph_debug3.py 2>&1 | grep -v amdgpu.ids ) >> gpurun_out/s10_debug3.log
( DBG_B=16 timeout 200 python benchmarks/x6_graph_debug3.py 2>&1 | grep -v amdgpu.ids ) >> gpurun_out/s10_debug3.log
( DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 python benchmarks/x6_graph_debug3.py 2>&1 | grep -v amdgpu.ids ) >> gpurun_out/s10_debug3.log
cat gpurun_out/s10_debug3.log
