#!/bin/bash
# HBM traffic of the bench step per kernel, from the L2 memory-side counters (two PMC passes: FETCH_SIZE and WRITE_SIZE
# do not fit one pass; counters + kernel trace only).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-vit}
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/hbm_${WL}_rd -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/hbm_${WL}_rd.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/hbm_${WL}_wr -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/hbm_${WL}_wr.log 2>&1
cd $R
python tools/hbm_traffic.py $WL | tee gpurun_out/hbm_${WL}.txt
rm -rf gpurun_out/hbm_${WL}_rd gpurun_out/hbm_${WL}_wr      # raw traces are scratch; gpurun merges back at most 64 MiB
