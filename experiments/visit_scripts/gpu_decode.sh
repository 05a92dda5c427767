#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_generation_gpu.py tests/test_f4_variants_gpu.py tests/test_decoder_kernels_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider -k "generation or greedy or padded or sample or kv_cache or decode or cached or append or vq_indices or f4" > gpurun_out/pytest_decode.log 2>&1
echo "decode parity rc=$? $(tail -1 gpurun_out/pytest_decode.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_decode.log | head -20
for rep in 1 2; do timeout 300 python tools/decode_bench.py 8 1024 32 2>&1 | tail -1; done | tee gpurun_out/decode_bench.txt
( cd ab/r02 && timeout 300 python tools/decode_bench.py 8 1024 32 2>&1 | tail -1 | sed 's/^/r02: /' ) | tee -a gpurun_out/decode_bench.txt
