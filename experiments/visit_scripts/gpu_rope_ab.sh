#!/bin/bash
# rope_bridge A/B of prebuilt libraries: digests + times, then the rope parity tests on the last one
set -u
cd "$(dirname "$0")/.."
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for v in "$@"; do
  cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
  echo "== $v"; timeout 120 python experiments/tools/rope_ab.py 2>&1 | grep -v amdgpu.ids | tail -3
done
timeout 200 python -m pytest tests/test_decoder_kernels_gpu.py -q -m gpu -x -k "rope" 2>&1 | tail -2
cp $keep libra_amd/lib/liblibra_hip.so
