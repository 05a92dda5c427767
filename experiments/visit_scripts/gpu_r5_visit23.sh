#!/bin/bash
# round-5 visit 23: the ViT leg (configs[1]) with the persistent vs the one-tile 256x256 GEMM
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2; do for v in np pers; do cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so; echo -n "$v "; timeout 200 python bench.py --workload vit --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], 'gemm_ms', r['gemm_ms_per_step'], r['achieved'])"; done; done | tee gpurun_out/v23_vit_leg.txt
cp $keep libra_amd/lib/liblibra_hip.so
