#!/bin/bash
# round-5 visit 15: persistent dK/dV with per-item re-derived lane constants: A/B and parity
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
ATTN_WHICH=bwd TEST_LIB=kvq1 ./tools/gpu_attn_ab.sh kvq0 kvq1 kvq1r0
