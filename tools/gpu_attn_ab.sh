#!/bin/bash
# Attention backward structure A/B inside one box visit: parity of structures $OLD / $NEW, then alternating timings.
set -u
mkdir -p gpurun_out
NEW=${NEW:-4}; OLD=${OLD:-3}
for v in $OLD $NEW; do
LIBRA_ATTN_DKV=$v timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention or attn" -p no:cacheprovider > gpurun_out/pytest_dkv$v.log 2>&1
echo "structure $v parity rc=$? $(tail -1 gpurun_out/pytest_dkv$v.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_dkv$v.log | head -20
done
for rep in 1 2; do for v in $OLD $NEW; do echo -n "structure $v: "; LIBRA_ATTN_DKV=$v timeout 120 python tools/attn_bench.py all 2>&1 | tail -1; done; done | tee gpurun_out/attn_ab$NEW.txt
