#!/bin/bash
# round-5 visit 6: CU mask layout probe, budget-only attention, dK/dV cycle trace, f2/f3 throughput
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 120 python tools/cu_mask_probe.py 2>/dev/null | tee gpurun_out/cu_mask_layout.txt
timeout 120 python - <<'P' 2>/dev/null | tee gpurun_out/cu_budget_only.txt
import sys; sys.path.insert(0, '.')
import torch, json
from libra_amd import kernels as K
exec(open('tools/cu_budget_probe.py').read().split("print(json.dumps({\"reserve\": 0")[0])
for b in (0, 248, 240, 224):
    K.set_cu_budget(b)
    print(json.dumps({"budget_only": b, **run()}), flush=True)
K.set_cu_budget(0)
P
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
cp ab/libs/dkvdbg.so libra_amd/lib/liblibra_hip.so
timeout 120 python tools/dkv_trace.py 2>/dev/null | tee gpurun_out/dkv_trace.txt
cp $keep libra_amd/lib/liblibra_hip.so
timeout 300 python tools/f2f3_bench.py 2>/dev/null | tee gpurun_out/f2f3_bench.txt
