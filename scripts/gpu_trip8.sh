#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_rules.py -m gpu -q -x -p no:cacheprovider -k "linear or golden or module" 2>&1 | tail -3 )
echo "=== BN64 persistent prefetchR ==="; ( TE_LINEAR_BN=64 timeout 120 python benchmarks/linear_bench.py --skip-peak 2>&1 | grep -v amdgpu.ids | head -9 )
echo "=== BN64 persistent no prefetch ==="; ( TE_LINEAR_BN=64 timeout 120 python benchmarks/linear_bench.py --skip-peak --lib benchmarks/libte_noprefetch.so 2>&1 | grep -v amdgpu.ids | head -9 )
echo "=== BN64 persistent abl4 (no epilogue memory) ==="; ( TE_LINEAR_BN=64 timeout 120 python benchmarks/linear_bench.py --skip-peak --lib benchmarks/libte_ablate4.so 2>&1 | grep -v amdgpu.ids | head -9 )
echo "=== BN64 non-persistent abl4 ==="; ( TE_LINEAR_PERSIST=0 TE_LINEAR_BN=64 timeout 120 python benchmarks/linear_bench.py --skip-peak --lib benchmarks/libte_ablate4.so 2>&1 | grep -v amdgpu.ids | head -9 )
