#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "splitk" -p no:cacheprovider > gpurun_out/v7_a.log 2>&1; echo "splitk tests rc=$? $(tail -1 gpurun_out/v7_a.log)"; grep -E "^E  |^FAILED" gpurun_out/v7_a.log | head
for i in 1 2; do
timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > gpurun_out/v7_head.json; python -c "
import json; d=json.load(open('gpurun_out/v7_head.json')); print('headline ms', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], d['roofline']['achieved'], d['step_check'])"
done
