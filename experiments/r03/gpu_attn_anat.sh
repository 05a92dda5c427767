#!/bin/bash
set -u
mkdir -p gpurun_out
out=gpurun_out/attn_fwd2_anatomy.txt; : > $out
for d in 0 1 2 3 4 5 6 0; do echo -n "dbg=$d " >> $out; LIBRA_ATTN_FWD=2 LIBRA_ATTN_DBG=$d timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out; done
echo -n "dbg=0 no out_lo " >> $out; ATTN_LO=0 LIBRA_ATTN_FWD=2 timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out
echo -n "v1 no out_lo " >> $out; ATTN_LO=0 LIBRA_ATTN_FWD=1 timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out
echo -n "v1 " >> $out; LIBRA_ATTN_FWD=1 timeout 120 python tools/attn_bench.py fwd 2>&1 | tail -1 >> $out
cat $out
