L=$PWD/transformer-explainability_amd/lib
mkdir -p gpurun_out/r06
for rep in 1 2; do
for v in old study; do
  echo "== $v (rep $rep)"
  TE_RELPROP_LIB=$L/libte_relprop_$v.so TE_X6_SNAP=1 python benchmarks/x6_variants.py --iters 10 --variants opt0 2>&1 | grep -E "TOTAL|Error|rror"
  TE_RELPROP_LIB=$L/libte_relprop_$v.so python benchmarks/x6_variants.py --iters 10 --variants opt0 2>&1 | grep -E "TOTAL|Error|rror"
done
done
timeout 900 python -m pytest tests/test_gpu_rules.py -q -m gpu -k "x6 or linear" -x 2>&1 | tail -3
