#!/bin/bash
# One GPU trip: parity tests, smoke, bench line, rocprofv3 kernel stats of the same command, streaming-kernel rates,
# the self-launching 2-rank path on the one-GPU rig.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh'      (logs land in gpurun_out/)
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl; rm -rf gpurun_out/prof; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 --maxfail=6 2>&1 | tail -60 ) > gpurun_out/tests_full.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
( timeout 400 python bench.py > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err )
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d "$OLDPWD/gpurun_out/prof" -o bench -- \
    python "$OLDPWD/bench.py" --cpu-baseline off > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof_bench.err" )
rm -f gpurun_out/prof/*agent_info* gpurun_out/prof/*kernel_trace*
( timeout 300 python bench.py --producers stock --cpu-baseline off > gpurun_out/bench_b64_stock.json 2> gpurun_out/bench_b64_stock.err )
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 500 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-maps 2 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err )
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d "$OLDPWD/gpurun_out/prof_$cfg" -o bench -- \
      python "$OLDPWD/bench.py" --config $cfg --steps 2 --warmup 1 --cpu-baseline off > /dev/null 2> "$OLDPWD/gpurun_out/prof_$cfg.err" )
  rm -f gpurun_out/prof_$cfg/*agent_info* gpurun_out/prof_$cfg/*kernel_trace*
done
# opt-in Linear rules on bf16 MFMAs: the three configurations' bench lines under the switch
( for cfg in vit_b16_224 vit_l16_384 bert_base_512; do
    TE_LINEAR_X6=1 timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline off --no-roofline 2>/dev/null | cut -c1-260
  done ) > gpurun_out/bench_linear_x6.log
( timeout 200 python scripts/stream_kernels_bw.py 2>&1 | tail -12 ) > gpurun_out/stream_kernels_bw.log
( TE_HEADMEAN_VARIANT=0 timeout 100 python scripts/stream_kernels_bw.py --only headmean 2>&1 | tail -3 ) >> gpurun_out/stream_kernels_bw.log
( for impl in rules tiles; do echo "TE_ATTN_IMPL=$impl"; for shape in "64 12 197" "32 16 577" "32 12 512"; do
    TE_ATTN_IMPL=$impl timeout 120 python scripts/attn_bench.py $shape 64 producers 2>&1 | grep -v amdgpu | tail -2; done; done ) > gpurun_out/attn_bench.log
( TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 300 python bench.py --gpus 2 --batch 16 --steps 2 --warmup 1 \
    > gpurun_out/bench_2rank_rig.json 2> gpurun_out/bench_2rank_rig.err )
echo "=== tests ==="; cat gpurun_out/tests_full.log
echo "=== smoke ==="; cat gpurun_out/smoke.log
echo "=== bench ==="; cat gpurun_out/bench_b64.json; tail -25 gpurun_out/bench_b64.err
echo "=== bench, stock producers ==="; cut -c1-330 gpurun_out/bench_b64_stock.json; tail -4 gpurun_out/bench_b64_stock.err
for cfg in vit_l16_384 bert_base_512; do echo "=== bench $cfg ==="; cut -c1-300 gpurun_out/bench_$cfg.json; tail -3 gpurun_out/bench_$cfg.err; done
echo "=== rocprof top kernels ==="; head -14 gpurun_out/prof/bench_kernel_stats.csv | cut -d, -f1-4
echo "=== bench lines with TE_LINEAR_X6=1 ==="; cat gpurun_out/bench_linear_x6.log
echo "=== streaming kernels ==="; cat gpurun_out/stream_kernels_bw.log
echo "=== attention rules ==="; cat gpurun_out/attn_bench.log
echo "=== 2-rank rig ==="; cut -c1-400 gpurun_out/bench_2rank_rig.json; tail -3 gpurun_out/bench_2rank_rig.err
