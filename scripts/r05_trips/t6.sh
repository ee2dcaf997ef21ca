#!/bin/bash
# trip 6: AV kb kernel with loads two tiles ahead and the stores first in a tile; ablations again
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention or einsum" 2>&1 | tail -5 ) > gpurun_out/t6_tests.log
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t6_build_study.log
for shape in "64 12 197 64" "32 12 512 64" "32 16 577 64"; do
  for st in old 0 1 2 4 5 0; do
    if [ $st = old ]; then export TE_ATTN_AV=old; else export TE_ATTN_AV=new TE_ATTN_KB_STUDY=$st; fi
    ( echo -n "study=$st "; timeout 120 python scripts/attn_bench.py $shape producers 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' '; echo ) >> gpurun_out/t6_av_study.log
  done
done
cat gpurun_out/t6_tests.log; cat gpurun_out/t6_av_study.log
