#!/bin/bash
# the shipped library after the study-only variants left it (fp32 kb AV kernel, direct GELU split): attention + GELU parity again
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 95 python -m pytest tests/test_gpu_rules.py tests/test_gpu_producers.py -m gpu -q -p no:cacheprovider -x -k "attention_rules or einsum or gelu or mlp_block or attention_backward_producer or attention_forward_producer" 2>&1 | tail -4 ) > gpurun_out/t43_tests.log
cat gpurun_out/t43_tests.log
