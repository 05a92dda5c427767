#!/bin/bash
# same-box sweep of prebuilt libraries over a GEMM shape set:  gpurun -- 'SWEEP_TILES=256,X experiments/visit_scripts/gpu_sweep_ab.sh text lib1 lib2'
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
setname=$1; shift
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for v in "$@"; do
  cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
  timeout 200 python tools/gemm_sweep.py $setname 10 > gpurun_out/sweep_$v.jsonl 2> gpurun_out/sweep_$v.txt
  echo "== $v"; grep -v "^\[run\]" gpurun_out/sweep_$v.txt | grep -v amdgpu.ids | tail -14
done
cp $keep libra_amd/lib/liblibra_hip.so
