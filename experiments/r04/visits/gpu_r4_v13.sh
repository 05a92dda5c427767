#!/bin/bash
# visit 13: parameters adopted into the fused operands - boundary / trainer / dp / generation / model parity, then the step A/B
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_generation_gpu.py tests/test_decoder_model_gpu.py tests/test_configs_gpu.py tests/test_f4_variants_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 600 bash experiments/visit_scripts/gpu_ab_step.sh
