#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_decoder_kernels_gpu.py tests/test_decoder_model_gpu.py tests/test_parity_fullsize_gpu.py tests/test_boundary_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_quick.log 2>&1
echo "quick parity rc=$? $(tail -1 gpurun_out/pytest_quick.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_quick.log | head -20
./experiments/visit_scripts/gpu_ab_step.sh
