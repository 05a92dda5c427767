#!/bin/bash
# A/B of the dK/dV pass: P shared by the (dV wave, dK wave) pair (LIBRA_ATTN_DKV=2) against both waves computing S
set -u
mkdir -p gpurun_out
for v in 1 2; do
LIBRA_ATTN_DKV=$v timeout 400 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -x -k "bridge_attention" -p no:cacheprovider > gpurun_out/pytest_dkv$v.log 2>&1
echo "dkv$v attention tests rc=$? $(tail -1 gpurun_out/pytest_dkv$v.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_dkv$v.log | head -10
done
for rep in 1 2; do
  for v in 1 2; do echo "== LIBRA_ATTN_DKV=$v"; LIBRA_ATTN_DKV=$v timeout 120 python tools/attn_bench.py all 2>&1 | grep -v amdgpu.ids | tail -4; done
done | tee gpurun_out/dkv_ab.txt
