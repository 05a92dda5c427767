#!/bin/bash
# round-5 visit 2: dQ pass in the unit / two-group structure (dq6) and the persistent dK/dV switch, A/B inside one box
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp libra_amd/lib/liblibra_hip.so ab/libs/_wt.so
# parity of the working tree first (a wrong kernel is not worth timing)
timeout 300 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -x -k "bridge_attention" -p no:cacheprovider > gpurun_out/v2_attn_pytest.log 2>&1
echo "attention tests (working tree) rc=$? $(tail -1 gpurun_out/v2_attn_pytest.log)"; grep -E "^E  |^FAILED" gpurun_out/v2_attn_pytest.log | head
out=gpurun_out/v2_attn_ab.txt; : > $out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2 3; do
  for v in v6 dkvp dq6_dkvp0 dq6_dkvp1; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    echo -n "$v " >> $out; timeout 90 python tools/attn_bench.py bwd 2>&1 | tail -1 >> $out
  done
done
cp $keep libra_amd/lib/liblibra_hip.so
cat $out
