#!/bin/bash
# timing anatomy of the lane-linear dK/dV kernel (DBG variants compute wrong results: timing only)
mkdir -p gpurun_out
for d in ${DBGS:-0 32 64 31}; do echo -n "dkv4 dbg $d: "; LIBRA_ATTN_DKV=4 LIBRA_DKV_DBG=$d timeout 120 python tools/attn_bench.py bwd 2>&1 | tail -1; done | tee gpurun_out/dkv4_anatomy.txt
