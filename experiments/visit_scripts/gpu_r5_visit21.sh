#!/bin/bash
# round-5 visit 21: persistent GEMM with the next tile's first K tile requested before the epilogue (gp2) vs plain persistent (gp1): GEMM parity on the
# persistent build, isolated shapes, then the headline step
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
cp ab/libs/gp2.so libra_amd/lib/liblibra_hip.so
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm -p no:cacheprovider 2>&1 | tail -3
for shape in "11760 22016 4096 0 0" "11760 4096 22016 0 1" "11760 12352 4096 0 0" "11760 4096 4096 0 0" "4624 11008 2752 0 0" "11008 2752 4672 1 1"; do
  for rep in 1 2; do for v in gp1 gp2; do cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so; echo -n "$v "; timeout 60 python tools/gemm_one.py $shape 20 2>&1 | tail -1; done; done
done | tee gpurun_out/v21_gemm_overlap.txt
cp $keep libra_amd/lib/liblibra_hip.so
./tools/gpu_lib_ab.sh 2 gp1 gp2
