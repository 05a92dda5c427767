#!/bin/bash
set -u
mkdir -p gpurun_out
for st in 3 4; do
LIBRA_ATTN_FWD=$st timeout 900 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/pytest_attn_fwd$st.log 2>&1
echo "structure $st parity rc=$? $(tail -1 gpurun_out/pytest_attn_fwd$st.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_attn_fwd$st.log | head -10
done
out=gpurun_out/attn_fwd3_t4.txt; : > $out
for rep in 1 2; do for st in 1 2 3 4; do LIBRA_ATTN_FWD=$st timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out; done; done
cat $out
