#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "splitk or skinny" -p no:cacheprovider > gpurun_out/v6_a.log 2>&1; echo "splitk tests rc=$? $(tail -1 gpurun_out/v6_a.log)"; grep -E "^E  |^FAILED" gpurun_out/v6_a.log | head
timeout 400 python -m pytest tests/test_decoder_model_gpu.py -m gpu -q -x -k "depth32" -p no:cacheprovider > gpurun_out/v6_b.log 2>&1; echo "depth32 tests rc=$? $(tail -1 gpurun_out/v6_b.log)"; grep -E "^E  |^FAILED" gpurun_out/v6_b.log | head
timeout 500 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "configs4" -p no:cacheprovider > gpurun_out/v6_c.log 2>&1; echo "cfg4 test rc=$? $(tail -1 gpurun_out/v6_c.log)"; grep -E "^E  |^FAILED" gpurun_out/v6_c.log | head
grep -E "depth 32|configs\[4\]" gpurun_out/parity_report.txt | tail -4 | cut -c1-900
timeout 200 python bench.py --seq 700 --with-optimizer --steps 6 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > gpurun_out/v6_cfg3.json; python -c "
import json; d=json.load(open('gpurun_out/v6_cfg3.json')); print('cfg3 ms', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], d['roofline']['achieved']); [print(x) for x in d['roofline']['by_shape'] if x['shape'].startswith('976')]"
