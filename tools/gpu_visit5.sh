#!/bin/bash
# Visit 5: f3 image pipeline tests, residual tests, backward structure 3 (lane-constant LDS addressing) parity + A/B,
# GEMM tile-structure choice on the shapes where the 256^2 kernel is weakest.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
for f in test_preprocess_gpu; do
  timeout 600 python -m pytest tests/$f.py -m gpu -q --maxfail=30 --timeout 500 -p no:cacheprovider -s > gpurun_out/pytest_$f.log 2>&1
  echo "$f rc=$? $(tail -1 gpurun_out/pytest_$f.log | cut -c1-200)"
done
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_decoder_kernels_gpu.py -m gpu -q -k "residual" -p no:cacheprovider -s > gpurun_out/pytest_residual.log 2>&1
echo "residual rc=$? $(tail -1 gpurun_out/pytest_residual.log)"; grep -h "rel err" gpurun_out/pytest_residual.log
LIBRA_ATTN_DKV=3 timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/pytest_dkv3.log 2>&1
echo "structure 3 parity rc=$? $(tail -1 gpurun_out/pytest_dkv3.log)"
for rep in 1 2; do for v in 1 3; do echo -n "bwd structure $v: "; LIBRA_ATTN_DKV=$v timeout 120 python tools/attn_bench.py bwd 2>&1 | tail -1; done; done | tee gpurun_out/attn_ab3.txt
for shape in "4624 11008 1024 0 0" "4624 4096 1024 0 1" "4624 1024 4096 0 0" "4624 1024 4096 0 1" "4624 4096 5504 0 0" "4624 4096 3136 0 0" "4624 1024 11008 0 1" "4616 1024 4096 0 1" "4616 4096 1024 0 1" "4616 3072 1024 0 1" "4616 1024 1024 0 1" "4624 5504 4096 0 1"; do
  for k in 256 128; do echo -n "kernel $k: "; LIBRA_GEMM_KERNEL=$k timeout 120 python tools/gemm_one.py $shape 30 2>&1 | tail -1; done
done | tee gpurun_out/gemm_choice.txt
