#!/bin/bash
# round 4, visit 1: parity of the three GEMM tile structures + the edited dK/dV pass, then the per-shape structure sweep
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" -p no:cacheprovider > gpurun_out/v1_pytest_gemm.log 2>&1
echo "gemm tests rc=$? $(tail -1 gpurun_out/v1_pytest_gemm.log)"
grep -E "^E  |^FAILED" gpurun_out/v1_pytest_gemm.log | head -20
timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/v1_pytest_attn.log 2>&1
echo "attention tests rc=$? $(tail -1 gpurun_out/v1_pytest_attn.log)"
grep -E "^E  |^FAILED" gpurun_out/v1_pytest_attn.log | head -20
timeout 600 python tools/gemm_sweep.py all 10 > gpurun_out/v1_gemm_sweep.jsonl 2> gpurun_out/v1_gemm_sweep.txt
echo "sweep rc=$?"; tail -4 gpurun_out/v1_gemm_sweep.txt
timeout 200 python tools/attn_bench.py all 2>&1 | tail -1 | tee gpurun_out/v1_attn_bench.txt
