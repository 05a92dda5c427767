#!/bin/bash
# Round-4 evidence visit A: HBM-traffic PMC passes of the headline workload FIRST (so the bench line of the same visit carries a
# non-stale roofline.traffic), then the driver's bench command, rocprof kernel stats, and the configs[3] / configs[4] shaped steps.
#   gpurun -- 'LIBRA_HEAD=<git rev-parse --short HEAD> experiments/visit_scripts/gpu_r4_final.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
./tools/hbm_traffic.sh libra > gpurun_out/hbm_libra.log 2>&1; tail -1 gpurun_out/hbm_libra.log | cut -c1-400
cp gpurun_out/hbm_libra.json profiles/r04_hbm_traffic_libra.json
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-700
bash tools/gpu_prof.sh 2>&1 | tail -2 | cut -c1-300
timeout 600 python bench.py --seq 700 --with-optimizer --no-extra --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_cfg3.log 2>&1; echo "cfg3 rc=$?"; tail -1 gpurun_out/bench_cfg3.log | cut -c1-300
timeout 900 python bench.py --seq 4096 --batch 2 --full-finetune --with-optimizer --recompute --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench_cfg4.log 2>&1; echo "cfg4 rc=$?"; tail -1 gpurun_out/bench_cfg4.log | cut -c1-300
