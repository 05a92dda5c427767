#!/bin/bash
# round-5 visit 3: dK/dV pass in the one-variant / register-operand / two-role structure (dkv6): parity, then A/B against the old pass
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp libra_amd/lib/liblibra_hip.so ab/libs/_wt.so
timeout 300 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention" -p no:cacheprovider > gpurun_out/v3_attn_pytest.log 2>&1
echo "attention tests (working tree) rc=$? $(tail -1 gpurun_out/v3_attn_pytest.log)"; grep -E "^E  |^FAILED" gpurun_out/v3_attn_pytest.log | head -30
out=gpurun_out/v3_attn_ab.txt; : > $out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2 3; do
  for v in dq6_dkvp0 _wt; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    echo -n "$v " >> $out; timeout 90 python tools/attn_bench.py bwd 2>&1 | tail -1 >> $out
  done
done
cp $keep libra_amd/lib/liblibra_hip.so
cat $out
