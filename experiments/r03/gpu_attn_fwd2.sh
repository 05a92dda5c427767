#!/bin/bash
# fwd2 vs fwd1: parity of both structures, then timings in separate processes (the structure is chosen at first call)
set -u
mkdir -p gpurun_out
for st in 2 1; do
  LIBRA_ATTN_FWD=$st timeout 900 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/pytest_attn_fwd$st.log 2>&1
  echo "structure $st parity rc=$? $(tail -1 gpurun_out/pytest_attn_fwd$st.log)"
  grep -E "^E  |^FAILED" gpurun_out/pytest_attn_fwd$st.log | head -20
done
for rep in 1 2 3; do for st in 2 1; do LIBRA_ATTN_FWD=$st timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1; done; done | tee gpurun_out/attn_fwd2_ab.txt
ATTN_S=700 LIBRA_ATTN_FWD=2 python tools/attn_bench.py fwd | tail -1 | tee -a gpurun_out/attn_fwd2_ab.txt
ATTN_S=700 LIBRA_ATTN_FWD=1 python tools/attn_bench.py fwd | tail -1 | tee -a gpurun_out/attn_fwd2_ab.txt
ATTN_S=4096 ATTN_B=2 LIBRA_ATTN_FWD=2 python tools/attn_bench.py fwd | tail -1 | tee -a gpurun_out/attn_fwd2_ab.txt
ATTN_S=4096 ATTN_B=2 LIBRA_ATTN_FWD=1 python tools/attn_bench.py fwd | tail -1 | tee -a gpurun_out/attn_fwd2_ab.txt
