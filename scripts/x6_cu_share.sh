#!/bin/bash
# Overlapped-step study (measurement build): the persistent x6 kernels on n of the 32 CUs of every XCD, the rest left to the
# memory-bound kernels of the other streams.   gpurun -- 'bash scripts/x6_cu_share.sh'
export TE_RELPROP_LIB=$PWD/transformer-explainability_amd/lib/libte_relprop_study.so
for n in 32 30 28 26 32; do
  TE_X6_CUS_PER_XCD=$n python bench.py --steps 12 --warmup 3 --cpu-baseline off --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('x6 on', $n, 'CUs per XCD:', round(d['value'],1), 'maps/s', round(d['ms_per_step'],2), 'ms')"
done
