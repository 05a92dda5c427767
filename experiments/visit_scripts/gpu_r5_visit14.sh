#!/bin/bash
# round-5 visit 14: persistent dK/dV (dkv6) with the rotation schedule: A/B (shipped / persistent / persistent without the row prefetch), parity on the persistent build
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
ATTN_WHICH=bwd TEST_LIB=kvp1 ./tools/gpu_attn_ab.sh kvp0 kvp1 kvp1r0
