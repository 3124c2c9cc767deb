#!/bin/bash
# GPU box: the whole -m gpu suite, full log under gpurun_out/, summary lines on stdout.   tools/gpu_tests.sh [tag]
cd "$(dirname "$0")/.."
T=${1:-run}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_$T.log 2>&1
echo "pytest rc $?"
grep -E "^FAILED|^ERROR| passed| failed" gpurun_out/pytest_$T.log | tail -15
