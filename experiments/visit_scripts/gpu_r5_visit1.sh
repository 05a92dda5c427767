#!/bin/bash
# round-5 visit 1: attention backward persistent A/B (ab/libs/v6.so vs the working tree), the new parity tests, a short bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp libra_amd/lib/liblibra_hip.so ab/libs/_wt.so
./tools/gpu_attn_ab.sh v6 _wt
timeout 900 python -m pytest tests/test_parity_fulldepth_gpu.py tests/test_parity_fullsize_gpu.py tests/test_decoder_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "full_depth or vq_indices or depth32_vs" > gpurun_out/v1_parity.log 2>&1
echo "parity rc=$? $(tail -1 gpurun_out/v1_parity.log)"; grep -E "^E  |^FAILED" gpurun_out/v1_parity.log | head -20
cat gpurun_out/parity_report.txt 2>/dev/null | tail -8
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v1_bench.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/v1_bench.log | cut -c1-400
