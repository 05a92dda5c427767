#!/bin/bash
# One GPU-box visit: probe, parity tests, smoke, bench, rocprof kernel stats. Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/probe_tr.bin > gpurun_out/probe_tr.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
nproc > gpurun_out/host.txt; lscpu | head -20 >> gpurun_out/host.txt
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1 ); echo "prof rc=$?"
find gpurun_out/prof -name "*stats*" | head
# HBM traffic of the GEMM launches (PMC)
./tools/hbm_traffic.sh vit > gpurun_out/hbm_vit.log 2>&1; tail -2 gpurun_out/hbm_vit.log | cut -c1-400
# full pretraining step (ViT+VQ -> Libra-11B decoder fwd+bwd): bench line + kernel stats, when asked for
if [ "${1:-}" = "libra" ]; then
  timeout 600 python bench.py --workload libra --steps 6 --warmup 2 > gpurun_out/bench_libra.log 2>&1; echo "bench libra rc=$?"; tail -1 gpurun_out/bench_libra.log | cut -c1-300
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_libra -- python $GRAFT_REPO_ROOT/bench.py --workload libra --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_libra.log 2>&1 ); echo "prof libra rc=$?"
fi
