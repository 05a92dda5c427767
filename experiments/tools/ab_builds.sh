#!/bin/bash
# A/B of two builds of the library inside ONE box visit (box-to-box variance, 3-15 %, is larger than most effects):
# build the variants into libra_amd/lib/{base,exp}.so.tmp (git-ignored, they travel with the gpurun snapshot), then
#   gpurun -- ./tools/ab_builds.sh
# alternates them under the same shapes, runs the GEMM parity tests on `exp`, and the bench on both.
L=libra_amd/lib
for shape in "2048 4096 1024 0 0" "2048 1024 4096 0 1" "18464 1024 1024 0 0" "4624 1024 1024 0 0"; do
  for rep in 1 2; do
    for v in base exp; do
      cp $L/$v.so.tmp $L/liblibra_hip.so
      echo -n "$v "; python tools/gemm_one.py $shape 30 2>&1 | tail -1
    done
  done
done
cp $L/exp.so.tmp $L/liblibra_hip.so
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm 2>&1 | tail -3
for rep in 1 2; do for v in base exp; do cp $L/$v.so.tmp $L/liblibra_hip.so; echo -n "$v "; python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200; done; done
