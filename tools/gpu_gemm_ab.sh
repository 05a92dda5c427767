#!/bin/bash
# GEMM A/B: parity of the variants, then alternating timings on the decoder's dominant shapes
mkdir -p gpurun_out
for v in ${VARIANTS:-1 2}; do
LIBRA_GEMM_DM=$v timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" -p no:cacheprovider > gpurun_out/pytest_gemm_dm$v.log 2>&1
echo "DM=$v parity rc=$? $(tail -1 gpurun_out/pytest_gemm_dm$v.log)"
done
for rep in 1 2; do for v in 0 ${VARIANTS:-1 2}; do echo "== LIBRA_GEMM_DM=$v"; LIBRA_GEMM_DM=$v timeout 200 python tools/gemm_big.py 10 2>&1 | tail -11; done; done > gpurun_out/gemm_dm_ab.txt
grep -E "==|sum|gate.up fwd|sq8k|K=1024" gpurun_out/gemm_dm_ab.txt
