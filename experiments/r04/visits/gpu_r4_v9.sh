#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -x -k "rope or cross_variants" -p no:cacheprovider > gpurun_out/v9_a.log 2>&1; echo "rope tests rc=$? $(tail -1 gpurun_out/v9_a.log)"; grep -E "^E  |^FAILED" gpurun_out/v9_a.log | head
./experiments/visit_scripts/gpu_ab_step.sh
