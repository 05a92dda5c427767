#!/bin/bash
# round-5 visit 10: dK/dV row prefetch A/B, call sites of the step's glue kernels
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
out=gpurun_out/v10_rowpre_ab.txt; : > $out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2 3; do
  for v in rowpre0 rowpre1; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    echo -n "$v " >> $out; timeout 90 python tools/attn_bench.py bwd 2>&1 | tail -1 >> $out
  done
done
cp ab/libs/rowpre1.so libra_amd/lib/liblibra_hip.so
timeout 200 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "bridge_attention_bwd" -p no:cacheprovider 2>&1 | tail -2
cp $keep libra_amd/lib/liblibra_hip.so
cat $out
timeout 300 python tools/glue_sites.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee gpurun_out/glue_sites.txt | head -80
