#!/bin/bash
# same-box A/B of prebuilt kernel libraries on the bridge-attention benchmark + parity of the last one (the working tree's build)
#   gpurun -- './tools/gpu_attn_ab.sh base pf1'
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
out=gpurun_out/attn_ab.txt; : > $out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2; do
  for v in "$@"; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    echo -n "$v " >> $out; timeout 90 python tools/attn_bench.py ${ATTN_WHICH:-all} 2>&1 | tail -1 >> $out
  done
done
cp $keep libra_amd/lib/liblibra_hip.so
cat $out
if [ -n "${TEST_LIB:-}" ]; then cp ab/libs/$TEST_LIB.so libra_amd/lib/liblibra_hip.so; echo "parity tests on ab/libs/$TEST_LIB.so"; fi
timeout 240 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/attn_ab_pytest.log 2>&1
echo "attention tests (working tree) rc=$? $(tail -1 gpurun_out/attn_ab_pytest.log)"
grep -E "^E  |^FAILED" gpurun_out/attn_ab_pytest.log | head
cp $keep libra_amd/lib/liblibra_hip.so
