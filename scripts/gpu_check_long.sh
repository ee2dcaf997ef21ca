#!/bin/bash
# Round 6, the long-sequence producers: producer tests, the same-box A/Bs, the two long configurations (bench lines + rocprofv3 kernel
# stats), PMC passes of the attention kernels at N = 197 and N = 577, rocprofv3 of the headline command (overlapped and serial).
#   gpurun --timeout 1500 -- 'bash scripts/gpu_check_long.sh'      (logs land in gpurun_out/)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
prof() {
  local out=$1; shift
  rm -rf gpurun_out/$out
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$ROOT/gpurun_out/$out" -o bench -- \
      python "$ROOT/bench.py" "$@" --cpu-baseline off > "$ROOT/gpurun_out/$out.json" 2> "$ROOT/gpurun_out/$out.err" )
  rm -f gpurun_out/$out/*agent_info* gpurun_out/$out/*kernel_trace*
}
( timeout 600 bash scripts/attn_fwd_long_ab.sh ) > gpurun_out/attn_fwd_long_ab.log 2>&1
( timeout 600 bash scripts/attn_bwd_long_ab.sh ) > gpurun_out/attn_bwd_long_ab.log 2>&1
for cfg in vit_l16_384 bert_base_512; do
  ( timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-maps 2 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err )
done
for cfg in vit_l16_384 bert_base_512; do prof prof_$cfg --config $cfg --steps 2 --warmup 1; done
( ATTN_PMC_SKIP_P2=1 timeout 250 bash scripts/attn_pmc.sh > gpurun_out/attn_pmc.log 2>&1 )
( ATTN_PMC_SKIP_P2=1 ATTN_PMC_SHAPE="32 16 577" ATTN_PMC_TAG=_n577 timeout 250 bash scripts/attn_pmc.sh > gpurun_out/attn_pmc_n577.log 2>&1 )
prof prof --steps 10
prof prof_serial --steps 10 --overlap-backward off --inflight 1
echo "=== fwd A/B ==="; head -3 gpurun_out/attn_fwd_long_ab.log; grep -E "^(new|w8|old)" gpurun_out/attn_fwd_long_ab.log | head -8
echo "=== bwd A/B ==="; head -3 gpurun_out/attn_bwd_long_ab.log; grep -E "^(new|old)" gpurun_out/attn_bwd_long_ab.log | head -6
for cfg in vit_l16_384 bert_base_512; do echo "=== bench $cfg ==="; cut -c1-330 gpurun_out/bench_$cfg.json; tail -2 gpurun_out/bench_$cfg.err; done
echo "=== attn pmc 577 ==="; grep -E "traffic|algorithmic|valu_per" gpurun_out/attn_pmc_summary_n577.csv
echo "=== headline under rocprof ==="; cut -c1-250 gpurun_out/prof.json
