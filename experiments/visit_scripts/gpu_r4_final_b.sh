#!/bin/bash
# Round-4 evidence visit B: the driver's other two commands (pytest -m gpu, smoke)
set -u
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=60 --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
