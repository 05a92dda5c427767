#!/bin/bash
# Attention kernels: parity, then timings (forward, backward) of the shipped structure.
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention or attn" -p no:cacheprovider > gpurun_out/pytest_attn.log 2>&1
echo "attention parity rc=$? $(tail -1 gpurun_out/pytest_attn.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_attn.log | head -20
for rep in 1 2; do timeout 120 python tools/attn_bench.py all 2>&1 | tail -1; done | tee gpurun_out/attn_bench.txt
