#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 200 python benchmarks/linear_bench.py --clock 2>&1 | tail -40 ) > gpurun_out/linbench_sw.log
( timeout 120 rocprofv3 -L 2>&1 | grep -i "SQ_\|GRBM\|TCC_EA0\|FETCH\|WRITE_SIZE" | cut -c1-160 | head -400 ) > gpurun_out/counters.txt
cd /tmp
( timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES -f csv -d /root/repo/gpurun_out/pmc1 -o p -- python /root/repo/benchmarks/linear_bench.py --skip-peak --reps 1 > /root/repo/gpurun_out/pmc1.log 2>&1 )
( timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU -f csv -d /root/repo/gpurun_out/pmc2 -o p -- python /root/repo/benchmarks/linear_bench.py --skip-peak --reps 1 > /root/repo/gpurun_out/pmc2.log 2>&1 )
cd /root/repo
echo "=== linbench ==="; cat gpurun_out/linbench_sw.log
echo "=== pmc1 ==="; tail -5 gpurun_out/pmc1.log; ls gpurun_out/pmc1
echo "=== pmc2 ==="; tail -5 gpurun_out/pmc2.log; ls gpurun_out/pmc2
wc -l gpurun_out/counters.txt
