#!/bin/bash
# visit 12: fused SwiGLU epilogue - kernel parity, model parity subset, then the same-process A/B
set -u
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "swiglu_pairs or every_tile" 2>&1 | tail -3
timeout 500 python -m pytest tests/test_decoder_model_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 400 python tools/swiglu_fusion_ab.py 3 6 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/swiglu_fusion_ab.txt
