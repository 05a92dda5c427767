#!/bin/bash
# round-5 visit 8: packed fp32 VALU phases (+ the -L log2e plane from the dQ pass): parity, then A/B against the previous build
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cp libra_amd/lib/liblibra_hip.so ab/libs/_wt.so
ATTN_WHICH=all ./tools/gpu_attn_ab.sh cur _wt
