#!/bin/bash
# trip 11: sweep50k on one GPU (20 global batches), the 2-rank gloo rig of it, two steps in flight (bounded waits now)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --config sweep50k --steps 20 --cpu-baseline off > gpurun_out/t11_sweep_1gpu.json 2> gpurun_out/t11_sweep_1gpu.err
tail -3 gpurun_out/t11_sweep_1gpu.err; python -c "
import json; d=json.loads(open('gpurun_out/t11_sweep_1gpu.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms/step', d['config']['global_batch'], d['config'].get('sweep_images'))"
TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 600 python bench.py --gpus 2 --config sweep50k --steps 6 --cpu-baseline off > gpurun_out/t11_sweep_rig2.json 2> gpurun_out/t11_sweep_rig2.err
tail -4 gpurun_out/t11_sweep_rig2.err; python -c "
import json; d=json.loads(open('gpurun_out/t11_sweep_rig2.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], d['n_gpus'], d.get('rig'), d['config']['global_batch'], d['config'].get('sweep_images'))"
TE_ALLOW_INFLIGHT=1 timeout 240 python bench.py --inflight 2 --steps 10 --cpu-baseline off > gpurun_out/t11_inflight2.json 2> gpurun_out/t11_inflight2.err
echo "inflight rc=$?"; tail -5 gpurun_out/t11_inflight2.err; python -c "
import json; d=json.loads(open('gpurun_out/t11_inflight2.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2))"
