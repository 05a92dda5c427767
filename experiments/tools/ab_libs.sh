#!/bin/bash
# Same-box A/B of two BUILDS of the kernel library in one gpurun visit (boxes differ by +-2..5 %, so numbers from different visits do
# not compare).  Build both variants here first - the library is deterministic, `cmp` tells whether a revert really is one:
#     make -C libra_amd/csrc && cp libra_amd/lib/liblibra_hip.so ab/libs/base.so      (then edit, rebuild, copy to ab/libs/<name>.so)
# ab/ is git-ignored but travels with the gpurun snapshot.  On the box:
#     gpurun -- './tools/ab_libs.sh "python tools/attn_bench.py all" base newidea'
# alternates the variants three times and prints the last line of the command for each.
set -u
cmd=$1; shift
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    echo -n "$v "; timeout 300 $cmd 2>&1 | tail -1
  done
done
cp $keep libra_amd/lib/liblibra_hip.so
