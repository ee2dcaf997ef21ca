#!/bin/bash
# One GPU trip (rounds 3-4): parity tests, smoke, the bench line, rocprofv3 kernel stats of the same command (and of the same
# step run serially: --overlap-backward off, what the roofline block's HIP-event times are comparable to), PMC passes of the
# x6 Linear kernels, the other two configurations, the self-launching 2-rank path on the one-GPU rig.
#   gpurun --timeout 1800 -- 'bash scripts/gpu_check.sh'      (logs land in gpurun_out/; SKIP_TESTS=1 skips pytest)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/prof_serial gpurun_out/prof_vit_l16_384 gpurun_out/prof_bert_base_512; export TMPDIR=/tmp; ROOT=$PWD
if [ -z "$SKIP_TESTS" ]; then
  rm -f gpurun_out/parity_report.jsonl
  ( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 --maxfail=6 2>&1 | tail -40 ) > gpurun_out/tests_full.log
fi
( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
( timeout 300 python bench.py --steps 20 > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err )
prof() {   # prof <outdir> <bench args...>: rocprofv3 kernel stats of one bench command, full kernel names (template arguments tell the x6 passes apart)
  local out=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$ROOT/gpurun_out/$out" -o bench -- \
      python "$ROOT/bench.py" "$@" --cpu-baseline off > "$ROOT/gpurun_out/$out.json" 2> "$ROOT/gpurun_out/$out.err" )
  rm -f gpurun_out/$out/*agent_info* gpurun_out/$out/*kernel_trace*
}
prof prof --steps 10
prof prof_serial --steps 10 --overlap-backward off --inflight 1
( timeout 400 bash scripts/x6_pmc.sh > gpurun_out/x6_pmc.log 2>&1 )
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-maps 2 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err )
  prof prof_$cfg --config $cfg --steps 2 --warmup 1
done
( TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 200 python bench.py --gpus 2 --batch 16 --steps 2 --warmup 1 --cpu-baseline off \
    > gpurun_out/bench_2rank_rig.json 2> gpurun_out/bench_2rank_rig.err )
( timeout 120 python scripts/stream_kernels_bw.py 2>&1 | tail -12 ) > gpurun_out/stream_kernels_bw.log
echo "=== tests ==="; tail -15 gpurun_out/tests_full.log
echo "=== smoke ==="; cat gpurun_out/smoke.log
echo "=== bench ==="; cut -c1-400 gpurun_out/bench_b64.json; tail -4 gpurun_out/bench_b64.err
for cfg in vit_l16_384 bert_base_512; do echo "=== bench $cfg ==="; cut -c1-300 gpurun_out/bench_$cfg.json; tail -3 gpurun_out/bench_$cfg.err; done
echo "=== rocprof top kernels (serial step) ==="; head -14 gpurun_out/prof_serial/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
echo "=== x6 pmc ==="; grep -E "traffic|hit_rate|valu_per" gpurun_out/x6_pmc_summary.csv
echo "=== 2-rank rig ==="; cut -c1-300 gpurun_out/bench_2rank_rig.json; tail -3 gpurun_out/bench_2rank_rig.err
