#!/bin/bash
# round-5 visit 17: forward with per-item re-derived lane constants (255 -> 231 VGPRs) vs before; parity on the working tree
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cp libra_amd/lib/liblibra_hip.so ab/libs/_wt.so
ATTN_WHICH=fwd ./tools/gpu_attn_ab.sh kvq1r0 _wt kvq1r0 _wt
