#!/bin/bash
# round-5 visit 16: dQ pass with per-item re-derived lane constants (spills 138 -> 67) vs before, both with the persistent dK/dV; parity on the working tree
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cp libra_amd/lib/liblibra_hip.so ab/libs/_wt.so
ATTN_WHICH=bwd ./tools/gpu_attn_ab.sh kvq1r0 _wt kvq1r0 _wt
