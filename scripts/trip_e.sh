#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
{
for d in 0 1 2 3 4 5; do
  echo "== TE_LSU_DIAG=$d"
  TE_LSU_DIAG=$d python benchmarks/linear_bench.py --skip-peak --reps 5 2>&1 | grep -E " cpass"
done
} 2>&1 | tee gpurun_out/trip_e.log
