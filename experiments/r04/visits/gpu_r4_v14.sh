#!/bin/bash
# visit 14: SGPR-base direct-to-LDS addressing in every GEMM structure and the ViT attention - kernel parity, then library A/B
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -3
tools/gpu_lib_ab.sh 2 g_base g_all
timeout 200 python bench.py --workload vit --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vit leg', d['value'], 'img/s', d['ms_per_step'], 'ms; gemm', d['roofline']['gemm_ms_per_step'], d['roofline']['achieved'])"
