#!/bin/bash
# full GPU suite + bench (state check after the attention kernels)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 --maxfail=6 2>&1 | tail -30 ) > gpurun_out/t27_tests.log
( timeout 300 python bench.py --steps 10 --cpu-baseline off > gpurun_out/t27_bench.json 2> gpurun_out/t27_bench.err )
echo "=== tests ==="; tail -12 gpurun_out/t27_tests.log
echo "=== bench ==="; cut -c1-300 gpurun_out/t27_bench.json; tail -3 gpurun_out/t27_bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/t27_bench.json"))
    for k in d["roofline"]["kernels"]:
        print(f'{k["name"]:28s} {k["avg_us"]:8.1f} us x{k["launches"]:3d}  frac {k["frac"]:.3f}')
except Exception as e:
    print("no roofline table:", e)
PY
