#!/bin/bash
# The GPU trips of a round end (rounds 3-5).  PART=A: parity tests, smoke, the bench line.  PART=B: rocprofv3 kernel stats of the
# same command (and of the same step run serially: --overlap-backward off, what the roofline block's HIP-event times are
# comparable to), the rig lines of the multi-rank path on the one GPU, the other two configurations, PMC passes of the x6
# Linear kernels and of the attention kernels, the GELU-plane A/B in the serial step, the lrp rule library.
#   gpurun --timeout 900 -- 'PART=A bash scripts/gpu_check.sh'      (logs land in gpurun_out/)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
PART=${PART:-AB}
prof() {   # prof <outdir> <bench args...>: rocprofv3 kernel stats of one bench command, full kernel names (template arguments tell the x6 passes apart)
  local out=$1; shift
  rm -rf gpurun_out/$out
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$ROOT/gpurun_out/$out" -o bench -- \
      python "$ROOT/bench.py" "$@" --cpu-baseline off > "$ROOT/gpurun_out/$out.json" 2> "$ROOT/gpurun_out/$out.err" )
  rm -f gpurun_out/$out/*agent_info* gpurun_out/$out/*kernel_trace*
}
if [[ $PART == *A* ]]; then
  rm -f gpurun_out/parity_report.jsonl
  ( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 --maxfail=6 2>&1 | tail -40 ) > gpurun_out/tests_full.log
  ( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
  ( timeout 300 python bench.py --steps 20 > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err )
  # the multi-rank path on the one GPU (rig: gloo, every rank on device 0), so that `--gpus 8` cannot rot between rounds:
  # the driver's own invocation shape (headline configuration, torch.distributed.run, 8 ranks) and the self-launched sweep
  ( TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      --master-port 29533 bench.py --gpus 8 --batch 8 --steps 2 --warmup 1 --cpu-baseline off > gpurun_out/bench_8rank_rig.json 2> gpurun_out/bench_8rank_rig.err )
  ( TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 300 python bench.py --gpus 8 --config sweep50k --batch 4 --steps 2 --warmup 1 --cpu-baseline off \
      > gpurun_out/bench_sweep50k_8rank_rig.json 2> gpurun_out/bench_sweep50k_8rank_rig.err )
  echo "=== tests ==="; tail -15 gpurun_out/tests_full.log
  echo "=== smoke ==="; cat gpurun_out/smoke.log
  echo "=== bench ==="; cut -c1-400 gpurun_out/bench_b64.json; tail -4 gpurun_out/bench_b64.err
  echo "=== 8-rank rig lines ==="; cut -c1-300 gpurun_out/bench_8rank_rig.json; tail -3 gpurun_out/bench_8rank_rig.err; cut -c1-300 gpurun_out/bench_sweep50k_8rank_rig.json; tail -3 gpurun_out/bench_sweep50k_8rank_rig.err
fi
if [[ $PART == *B* ]]; then
  if [ -n "$RERUN_TESTS" ]; then      # (a second look at tests changed after part A)
    ( timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$RERUN_TESTS" 2>&1 | tail -15 ) > gpurun_out/tests_rerun.log
    echo "=== tests (rerun: $RERUN_TESTS) ==="; tail -8 gpurun_out/tests_rerun.log
  fi
  prof prof --steps 10
  prof prof_serial --steps 10 --overlap-backward off --inflight 1
  ( TE_DIST_BACKEND=gloo TE_DEVICE_OVERRIDE=0 timeout 200 python bench.py --gpus 2 --batch 16 --steps 2 --warmup 1 --cpu-baseline off \
      > gpurun_out/bench_2rank_rig.json 2> gpurun_out/bench_2rank_rig.err )
  for cfg in vit_l16_384 bert_base_512; do
    ( timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-maps 2 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err )
  done
  ( timeout 400 bash scripts/x6_pmc.sh > gpurun_out/x6_pmc.log 2>&1 )
  ( ATTN_PMC_SKIP_P2=1 timeout 200 bash scripts/attn_pmc.sh > gpurun_out/attn_pmc.log 2>&1 )
  for f in 0 1; do
    ( echo -n "serial step, TE_X6_FUSE_GELU=$f: "; TE_X6_FUSE_GELU=$f timeout 200 python bench.py --steps 10 --overlap-backward off --inflight 1 --cpu-baseline off --no-roofline 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'maps/s', d['ms_per_step'], 'ms per step')" ) >> gpurun_out/gelu_planes_serial_ab.log
  done
  ( timeout 200 python bench.py --rules lrp --steps 10 --cpu-baseline off > gpurun_out/bench_b64_rules_lrp.json 2> gpurun_out/bench_b64_rules_lrp.err )
  for cfg in vit_l16_384 bert_base_512; do prof prof_$cfg --config $cfg --steps 2 --warmup 1; done
  ( timeout 120 python scripts/stream_kernels_bw.py 2>&1 | tail -12 ) > gpurun_out/stream_kernels_bw.log
  for cfg in vit_l16_384 bert_base_512; do echo "=== bench $cfg ==="; cut -c1-300 gpurun_out/bench_$cfg.json; tail -3 gpurun_out/bench_$cfg.err; done
  echo "=== rocprof top kernels (serial step) ==="; head -14 gpurun_out/prof_serial/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
  echo "=== x6 pmc ==="; grep -E "traffic|hit_rate|valu_per" gpurun_out/x6_pmc_summary.csv
  echo "=== attn pmc ==="; tail -12 gpurun_out/attn_pmc.log
  echo "=== rig lines ==="; cut -c1-300 gpurun_out/bench_sweep50k_8rank_rig.json; tail -3 gpurun_out/bench_sweep50k_8rank_rig.err; cut -c1-300 gpurun_out/bench_2rank_rig.json; tail -3 gpurun_out/bench_2rank_rig.err
  echo "=== gelu planes, serial step ==="; cat gpurun_out/gelu_planes_serial_ab.log
  echo "=== rules lrp ==="; cut -c1-200 gpurun_out/bench_b64_rules_lrp.json
fi
