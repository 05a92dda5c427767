#!/bin/bash
# Round-end visit: the driver's three commands (pytest -m gpu, smoke, bench) + rocprof kernel stats and HBM-traffic PMC passes of the
# headline workload + the configs[3] / configs[4] shaped steps.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_report.txt
[ "${SKIP_PYTEST:-0}" = 1 ] && PYT="tests/test_parity_fullsize_gpu.py" || PYT="tests"
timeout 1500 python -m pytest $PYT -m gpu -q --maxfail=60 --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-600
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bridge -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_bridge.log 2>&1 ); echo "prof rc=$?"
cp $(find gpurun_out/prof_bridge -name "*kernel_stats.csv" | head -1) gpurun_out/prof_bridge_kernel_stats.csv; rm -rf gpurun_out/prof_bridge
timeout 600 python bench.py --seq 700 --with-optimizer --no-extra --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_cfg3.log 2>&1; echo "cfg3 rc=$?"; tail -1 gpurun_out/bench_cfg3.log | cut -c1-400
timeout 900 python bench.py --seq 4096 --batch 2 --full-finetune --with-optimizer --recompute --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench_cfg4.log 2>&1; echo "cfg4 rc=$?"; tail -1 gpurun_out/bench_cfg4.log | cut -c1-400
./tools/hbm_traffic.sh libra > gpurun_out/hbm_libra.log 2>&1; tail -1 gpurun_out/hbm_libra.log | cut -c1-300
