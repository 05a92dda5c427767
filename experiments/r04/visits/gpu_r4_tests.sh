#!/bin/bash
# the whole -m gpu suite, bounded; tail + failures to gpurun_out/
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? $(tail -1 gpurun_out/pytest_gpu.log)"
grep -E "^E  |^FAILED|Error" gpurun_out/pytest_gpu.log | head -20
tail -5 gpurun_out/pytest_gpu.log > gpurun_out/pytest_gpu_tail.txt
