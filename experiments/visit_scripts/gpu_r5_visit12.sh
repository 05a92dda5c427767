#!/bin/bash
# round-5 visit 12: per-workgroup records of the forward (prologue anatomy)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
cp ab/libs/fwddbg256.so libra_amd/lib/liblibra_hip.so
timeout 120 python tools/attn_wg_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fwd_wg_times.txt
cp $keep libra_amd/lib/liblibra_hip.so
