#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( TE_BUILD_DEFINES=TE_STUDY timeout 600 python transformer-explainability_amd/build.py 2>&1 | tail -2 ) > gpurun_out/t29_build_study.log
for cfg in "old old" "x6 old" "old x6" "fp32kb old"; do
  set -- $cfg
  ( echo "AV=$1 QK=$2"; TE_ATTN_AV=$1 TE_ATTN_QK=$2 timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -x -k "test_config1_vit_b16_batch64" 2>&1 | tail -2 ) >> gpurun_out/t29.log
done
cat gpurun_out/t29.log
