#!/bin/bash
# One GPU trip (round 3): parity tests, smoke, the bench line, rocprofv3 kernel stats of the same command, PMC passes of
# the x6 Linear kernels, the other two configurations, the self-launching 2-rank path on the one-GPU rig.
#   gpurun --timeout 1800 -- 'bash scripts/gpu_check.sh'      (logs land in gpurun_out/)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl; rm -rf gpurun_out/prof gpurun_out/prof_vit_l16_384 gpurun_out/prof_bert_base_512; export TMPDIR=/tmp; ROOT=$PWD
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 --maxfail=6 2>&1 | tail -40 ) > gpurun_out/tests_full.log
( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
( timeout 300 python bench.py --steps 20 > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d "$ROOT/gpurun_out/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --cpu-baseline off > "$ROOT/gpurun_out/prof_bench.json" 2> "$ROOT/gpurun_out/prof_bench.err" )
rm -f gpurun_out/prof/*agent_info* gpurun_out/prof/*kernel_trace*
( timeout 200 python bench.py --linear fp32 --cpu-baseline off --steps 10 > gpurun_out/bench_b64_fp32.json 2> gpurun_out/bench_b64_fp32.err )
( timeout 400 bash scripts/x6_pmc.sh > gpurun_out/x6_pmc.log 2>&1 )
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-maps 2 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -f csv -d "$ROOT/gpurun_out/prof_$cfg" -o bench -- \
      python "$ROOT/bench.py" --config $cfg --steps 2 --warmup 1 --cpu-baseline off > /dev/null 2> "$ROOT/gpurun_out/prof_$cfg.err" )
  rm -f gpurun_out/prof_$cfg/*agent_info* gpurun_out/prof_$cfg/*kernel_trace*
done
( TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 200 python bench.py --gpus 2 --batch 16 --steps 2 --warmup 1 --cpu-baseline off \
    > gpurun_out/bench_2rank_rig.json 2> gpurun_out/bench_2rank_rig.err )
( timeout 120 python scripts/stream_kernels_bw.py 2>&1 | tail -12 ) > gpurun_out/stream_kernels_bw.log
echo "=== tests ==="; tail -15 gpurun_out/tests_full.log
echo "=== smoke ==="; cat gpurun_out/smoke.log
echo "=== bench ==="; cut -c1-400 gpurun_out/bench_b64.json; tail -4 gpurun_out/bench_b64.err
echo "=== bench, fp32-MFMA Linear rules ==="; cut -c1-200 gpurun_out/bench_b64_fp32.json
for cfg in vit_l16_384 bert_base_512; do echo "=== bench $cfg ==="; cut -c1-300 gpurun_out/bench_$cfg.json; tail -3 gpurun_out/bench_$cfg.err; done
echo "=== rocprof top kernels ==="; head -14 gpurun_out/prof/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
echo "=== x6 pmc ==="; grep -E "traffic|hit_rate|valu_per" gpurun_out/x6_pmc_summary.csv
echo "=== 2-rank rig ==="; cut -c1-300 gpurun_out/bench_2rank_rig.json; tail -3 gpurun_out/bench_2rank_rig.err
