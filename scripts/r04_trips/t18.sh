#!/bin/bash
# trip 18: robustness -- the model-level tests with EVERY x6 launch on a 16-workgroup grid (TE_X6_FLAGS=0x4000: nearly every tile
# is cut and handed over between workgroups), and with every launch pinned to each tile geometry in turn
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TE_X6_FLAGS=0x4000 timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "vit_b16 or config1 or bert_base or orig_lrp" > gpurun_out/t18_small_grid.log 2>&1
echo "small grid:"; grep -v amdgpu gpurun_out/t18_small_grid.log | tail -4
for pin in 1 2 3; do
  TE_X6_FLAGS=$pin timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "vit_b16_linear_x6_path or config1 or orig_lrp or bert_base_golden" > gpurun_out/t18_pin$pin.log 2>&1
  echo "pin $pin:"; grep -v amdgpu gpurun_out/t18_pin$pin.log | tail -2
done
