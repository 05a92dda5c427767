#!/bin/bash
# One GPU-box visit (round 2): parity tests, smoke, the driver's bench command, rocprof kernel stats of exactly that
# workload.  Everything lands in gpurun_out/.   usage: gpurun -- ./experiments/visit_scripts/gpu_visit.sh [tests|notests] [extra pytest args]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_report.txt
nproc > gpurun_out/host.txt; lscpu | head -20 >> gpurun_out/host.txt
if [ "${1:-tests}" = "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=60 --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -25 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
fi
timeout 1200 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -4 gpurun_out/bench.log | cut -c1-1500
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bridge -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_bridge.log 2>&1 ); echo "prof rc=$?"
find gpurun_out/prof_bridge -name "*kernel_stats*" | head -3
