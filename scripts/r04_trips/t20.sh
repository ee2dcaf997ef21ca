#!/bin/bash
# trip 20: the driver's default invocation after the comparison run went back to one step in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 240 python bench.py > gpurun_out/t20_default.json 2> gpurun_out/t20_default.err
echo "rc=$?"; grep -E "timed|comparison|captured" gpurun_out/t20_default.err
python -c "
import json; d=json.loads(open('gpurun_out/t20_default.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), d['steps'], d['config']['steps_in_flight'], d['config'].get('fp32_mfma_maps_per_s'), d['roofline']['frac'], d['cpu_baseline']['value'])"
