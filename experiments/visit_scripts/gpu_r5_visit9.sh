#!/bin/bash
# round-5 visit 9: the whole GPU test suite + smoke on the working tree, HBM-side traffic of both bench workloads (PMC, two passes each)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
grep -E "^E  |^FAILED" gpurun_out/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
./tools/hbm_traffic.sh libra | tail -12
./tools/hbm_traffic.sh vit | tail -8
