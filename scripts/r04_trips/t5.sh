#!/bin/bash
# trip 5: A/B of the library with and without the K-segment split code (same box, interleaved): is the un-split code slower?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for lib in libte_relprop.so libte_relprop_noksplit.so; do
    TE_RELPROP_LIB=$PWD/transformer-explainability_amd/lib/$lib timeout 300 python benchmarks/x6_variants.py --iters 8 --variants base,g128,g256 > gpurun_out/t5_${lib}_$rep.log 2>&1
    echo "== $lib rep $rep"; grep -E "^TOTAL" gpurun_out/t5_${lib}_$rep.log
  done
done
grep -h "VAR" gpurun_out/t5_libte_relprop.so_2.log | grep -E '"(base|g256)"' | cut -c1-220
echo; grep -h "VAR" gpurun_out/t5_libte_relprop_noksplit.so_2.log | grep -E '"(base|g256)"' | cut -c1-220
