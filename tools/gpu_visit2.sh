#!/bin/bash
# Visit 2: re-run the tests that failed in visit 1 + the new generation / DP tests, A/B the two dK/dV structures
# (parity + timing inside one box visit), GEMM kernel choice on the K=1024 shapes, then the driver's bench command.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_parity_fullsize_gpu.py tests/test_generation_gpu.py tests/test_dp_gpu.py tests/test_decoder_model_gpu.py -m gpu -q --maxfail=30 --timeout 600 -p no:cacheprovider > gpurun_out/pytest_v2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_v2.log; tail -12 gpurun_out/pytest_v2.log | cut -c1-300
# dK/dV structure 2: parity, then timing of both (alternating)
LIBRA_ATTN_DKV=2 timeout 600 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention_bwd" -p no:cacheprovider > gpurun_out/pytest_dkv2.log 2>&1
echo "dkv2 parity rc=$?"; tail -4 gpurun_out/pytest_dkv2.log | cut -c1-300
for rep in 1 2; do for v in 1 2; do echo -n "dkv structure $v: "; LIBRA_ATTN_DKV=$v timeout 120 python tools/attn_bench.py bwd 2>&1 | tail -1; done; done | tee gpurun_out/attn_ab.txt
# GEMM structure on the ViT's K = 1024 shapes
for k in 256 128; do for shape in "18464 4096 1024 0 0" "18464 4096 1024 0 1" "18464 1024 1024 0 0"; do echo -n "kernel $k: "; LIBRA_GEMM_KERNEL=$k timeout 120 python tools/gemm_one.py $shape 30 2>&1 | tail -1; done; done | tee gpurun_out/gemm_ab.txt
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-3000
