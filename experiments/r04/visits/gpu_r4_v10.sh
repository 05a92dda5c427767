#!/bin/bash
# visit 10: decoder/model parity subset after moving the loss counts onto the forward's one host read, then the same-box step A/B
set -u
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_decoder_model_gpu.py tests/test_f4_variants_gpu.py tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -3
sed -i 's/for rep in 1; do/for rep in 1 2; do/' experiments/visit_scripts/gpu_ab_step.sh
timeout 600 bash experiments/visit_scripts/gpu_ab_step.sh
