#!/bin/bash
# bench lines of the two other configurations with the final defaults (relprop beside the backward pass, one probe step)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 300 python bench.py --config $cfg --steps 6 --warmup 1 --cpu-maps 2 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err )
  cut -c1-330 gpurun_out/bench_$cfg.json; tail -2 gpurun_out/bench_$cfg.err
done
