#!/bin/bash
# round-5 visit 7: dK/dV cycle trace, CU budget probe (mask = highest bits, persistent grid = SE-symmetric part)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
cp ab/libs/dkvdbg.so libra_amd/lib/liblibra_hip.so
timeout 120 python tools/dkv_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dkv_trace.txt
cp $keep libra_amd/lib/liblibra_hip.so
timeout 200 python tools/cu_budget_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cu_budget_probe.txt
