#!/bin/bash
# the three configuration tests with the final end-to-end criteria (pooled reference distribution)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( timeout 420 python -m pytest tests -m gpu -q -p no:cacheprovider -k "test_config1 or test_config2 or test_config3" 2>&1 | tail -12 ) > gpurun_out/tests_rerun2.log
cat gpurun_out/tests_rerun2.log
