#!/bin/bash
# experiment: share of the 256^2 GEMM epilogue (LIBRA_DBG_EPI: 0 full, 1 none, 2 no global stores)
for shape in "16384 4096 1024" "16384 1024 1024" "16384 1024 4096" "16384 4096 4096"; do
  for rep in 1 2; do
    for m in 0 1 2; do
      echo -n "epi=$m "; LIBRA_DBG_EPI=$m python tools/gemm_one.py $shape 0 0 30 2>&1 | tail -1
    done
  done
done
