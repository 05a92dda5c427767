#!/bin/bash
# visit 11: parity after the sync-free assembly + device-side loss scale, then the same-box step A/B against the round-3 tree
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py tests/test_decoder_model_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 bash experiments/visit_scripts/gpu_ab_step.sh
