#!/bin/bash
# same-box A/B of the headline step: the previous round's tree (ab/r03 = git archive of 464dca6, built in place) against the working tree, alternating
set -u
mkdir -p gpurun_out
out=gpurun_out/ab_step.txt; : > $out
for rep in 1; do
  for v in r03 head; do
    if [ $v = r03 ]; then d=ab/r03; else d=.; fi
    ( cd $d && timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', 'ms_per_step', d['ms_per_step'], 'gemm_ms', d['roofline']['gemm_ms_per_step'], 'gemm_TF', d['roofline']['achieved'], 'launches', d['roofline']['launches'])" ) >> $out
  done
done
cat $out
