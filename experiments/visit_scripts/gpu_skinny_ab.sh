#!/bin/bash
# skinny-GEMM A/B of prebuilt libraries (digests + times), then the skinny / generation tests and the decode bench on the last one
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for v in "$@"; do
  cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
  echo "== $v"; timeout 120 python experiments/tools/skinny_ab.py 2>&1 | grep -v amdgpu.ids | tail -8
  timeout 200 python tools/decode_bench.py 8 1024 32 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done
timeout 300 python -m pytest tests/test_decoder_kernels_gpu.py tests/test_generation_gpu.py -q -m gpu -x -k "skinny or generat or decode or cached" 2>&1 | tail -2
cp $keep libra_amd/lib/liblibra_hip.so
