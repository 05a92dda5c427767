#!/bin/bash
# same-box A/B of prebuilt kernel libraries (ab/libs/<name>.so) on the headline step, alternating:  gpurun -- 'tools/gpu_lib_ab.sh 3 old new'
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
reps=$1; shift
out=gpurun_out/lib_ab.txt; : > $out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in $(seq $reps); do
  for v in "$@"; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
big=sum(s['ms'] for s in r['by_shape'] if s['shape'].startswith('11760'))
print('$v', 'ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'gemm_TF', r['achieved'], 'text_shapes_ms', round(big,2))" >> $out
  done
done
cp $keep libra_amd/lib/liblibra_hip.so
cat $out
