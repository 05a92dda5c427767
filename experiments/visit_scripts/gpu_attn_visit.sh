#!/bin/bash
# one GPU visit for an attention change: parity of the working tree's library first, then a same-box A/B of prebuilt libraries
#   gpurun -- './experiments/visit_scripts/gpu_attn_visit.sh fwd base v3 v3np'      (first argument: fwd | bwd | all)
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
which=$1; shift
timeout 400 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -x -k "bridge_attention" -p no:cacheprovider > gpurun_out/attn_pytest.log 2>&1
rc=$?
echo "attention tests (working tree) rc=$rc $(tail -1 gpurun_out/attn_pytest.log)"
grep -E "^E  |^FAILED" gpurun_out/attn_pytest.log | head -20
if [ $rc -eq 124 ]; then echo "TIMEOUT: not benchmarking"; exit 1; fi
out=gpurun_out/attn_ab.txt; : > $out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    echo -n "$v " >> $out; ATTN_WHICH=$which timeout 90 python tools/attn_bench.py $which 2>&1 | tail -1 >> $out
  done
done
cp $keep libra_amd/lib/liblibra_hip.so
cat $out
