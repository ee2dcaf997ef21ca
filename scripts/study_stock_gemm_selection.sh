#!/bin/bash
# stock-GEMM selection study: default vs rocBLAS-preferred vs TunableOp-tuned, on the bench step.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
show() { tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('maps/s %.1f  ms/step %.2f  cpass %.1f TF frac %.3f  zfwd %.1f TF' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['zpass']['achieved']))
except Exception as e: print('FAILED', e)"; }
{
echo "== default"
python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>/dev/null | show
echo "== TORCH_BLAS_PREFER_HIPBLASLT=0"
TORCH_BLAS_PREFER_HIPBLASLT=0 python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>/dev/null | show
echo "== TORCH_BLAS_PREFER_HIPBLASLT=1"
TORCH_BLAS_PREFER_HIPBLASLT=1 python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>/dev/null | show
echo "== TunableOp tuning run"
export PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_results.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
( time PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 timeout 420 python bench.py --steps 3 --warmup 2 --cpu-baseline off --graph off 2>gpurun_out/tunable_tuning.err | show ) 2>&1 | grep -v "^$" | grep -v user | grep -v sys
ls -la gpurun_out/tunableop_results*.csv 2>/dev/null; wc -l gpurun_out/tunableop_results*.csv 2>/dev/null
echo "== TunableOp tuned (no further tuning)"
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>/dev/null | show
} 2>&1 | tee gpurun_out/trip_c.log
