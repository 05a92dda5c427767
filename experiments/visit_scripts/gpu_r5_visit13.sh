#!/bin/bash
# round-5 visit 13: forward with the next item's prologue requested before the epilogue: parity (incl. the multi-item cases), A/B
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cp libra_amd/lib/liblibra_hip.so ab/libs/_wt.so
ATTN_WHICH=fwd ./tools/gpu_attn_ab.sh cur _wt cur _wt
