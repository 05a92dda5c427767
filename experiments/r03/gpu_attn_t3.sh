#!/bin/bash
set -u
mkdir -p gpurun_out
LIBRA_ATTN_FWD=2 timeout 900 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/pytest_attn_fwd2.log 2>&1
echo "structure 2 parity rc=$? $(tail -1 gpurun_out/pytest_attn_fwd2.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_attn_fwd2.log | head -20
out=gpurun_out/attn_fwd2_t3.txt; : > $out
for rep in 1 2; do for d in 0 10; do echo -n "dbg=$d " >> $out; LIBRA_ATTN_FWD=2 LIBRA_ATTN_DBG=$d timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out; done; done
for d in 3 4 6; do echo -n "dbg=$d " >> $out; LIBRA_ATTN_FWD=2 LIBRA_ATTN_DBG=$d timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out; done
echo -n "v1 " >> $out; LIBRA_ATTN_FWD=1 timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out
cat $out
