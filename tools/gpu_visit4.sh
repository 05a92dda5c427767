#!/bin/bash
# Visit 4: output-residual (out_lo) parity, VQ decoder tests, full-size parity, dkv structure A/B (scratch-free loop).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
for f in test_kernels_gpu test_decoder_kernels_gpu test_vq_decode_gpu test_model_gpu test_parity_fullsize_gpu test_decoder_model_gpu; do
  timeout 900 python -m pytest tests/$f.py -m gpu -q --maxfail=30 --timeout 600 -p no:cacheprovider -s > gpurun_out/pytest_$f.log 2>&1
  echo "$f rc=$? $(tail -1 gpurun_out/pytest_$f.log | cut -c1-200)"
done
grep -h "error .*->\|rel err" gpurun_out/pytest_test_kernels_gpu.log gpurun_out/pytest_test_decoder_kernels_gpu.log | head
LIBRA_ATTN_DKV=2 timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention_bwd" -p no:cacheprovider > gpurun_out/pytest_dkv2.log 2>&1
echo "dkv2 parity rc=$? $(tail -1 gpurun_out/pytest_dkv2.log)"
for rep in 1 2; do for v in 1 2; do echo -n "dkv structure $v: "; LIBRA_ATTN_DKV=$v timeout 120 python tools/attn_bench.py bwd 2>&1 | tail -1; done; done | tee gpurun_out/attn_ab.txt
