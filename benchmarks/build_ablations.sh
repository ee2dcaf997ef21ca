#!/bin/bash
# Measurement builds of the Linear.relprop kernels with parts of the main loop removed (TE_ABLATION, see
# csrc/te_linear.hip).  Output: benchmarks/libte_ablate{1,2,3}.so exporting the same C ABI (linear entry points only
# are meaningful).  Never shipped / never loaded by the package.
set -e
cd "$(dirname "$0")/.."
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
SRC=transformer-explainability_amd/csrc
for a in ${ABLATIONS:-1 2 3}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -DTE_ABLATION=$a ${EXTRA_DEFS} -I include -I $SRC -shared $SRC/te_api.hip $SRC/te_elementwise.hip $SRC/te_linear.hip $SRC/te_attn.hip \
    $SRC/te_attn_mfma.hip $SRC/te_rollout.hip -o benchmarks/libte_ablate$a.so -L $TL -Wl,-rpath,$TL 2>&1 | grep -i "error" || true ) &
done
wait
ls -la benchmarks/*.so
