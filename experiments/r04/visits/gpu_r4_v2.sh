#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python tools/gemm_sweep.py all 10 > gpurun_out/v2_gemm_sweep.jsonl 2> gpurun_out/v2_gemm_sweep.txt
echo "sweep rc=$?"; tail -12 gpurun_out/v2_gemm_sweep.txt
timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/v2_pytest_attn.log 2>&1
echo "attention tests rc=$? $(tail -1 gpurun_out/v2_pytest_attn.log)"
for i in 1 2; do timeout 200 python tools/attn_bench.py all 2>&1 | tail -1 | tee -a gpurun_out/v2_attn_bench.txt; done
