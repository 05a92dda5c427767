#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" -p no:cacheprovider > gpurun_out/v3_pytest_gemm.log 2>&1
echo "gemm tests rc=$? $(tail -1 gpurun_out/v3_pytest_gemm.log)"
grep -E "^E  |^FAILED" gpurun_out/v3_pytest_gemm.log | head -20
timeout 900 python tools/gemm_sweep.py all 10 > gpurun_out/v3_gemm_sweep.jsonl 2> gpurun_out/v3_gemm_sweep.txt
echo "sweep rc=$?"; grep -v "^\[run\]" gpurun_out/v3_gemm_sweep.txt | grep "ms/step"
