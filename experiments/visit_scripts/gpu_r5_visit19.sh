#!/bin/bash
# round-5 visit 19: in-step check of the forward's per-item lane constants (231 vs 255 VGPRs), dQ fragment-ring depth
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
./tools/gpu_lib_ab.sh 2 fwd255 head
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2; do for v in dqnf4 dqnf5 dqnf6; do cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so; echo -n "$v "; timeout 90 python tools/attn_bench.py bwd 2>&1 | tail -1; done; done | tee gpurun_out/v19_dqnf.txt
cp $keep libra_amd/lib/liblibra_hip.so
