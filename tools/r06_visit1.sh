#!/bin/bash
# Round-6 first GPU visit: parity of the multi-problem GEMM, its isolated timing, then the headline step with / without it
# (same box, alternating).  Everything lands in gpurun_out/r06_v1_*.
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
O=gpurun_out
(timeout 900 python -m pytest tests/test_zz_gemm_multi_gpu.py -x -q 2>&1 | tail -25) > $O/r06_v1_t_multi.log
(timeout 600 python tools/gemm_multi_bench.py 10 2>&1 | tail -20) > $O/r06_v1_multi_bench.log
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k gemm 2>&1 | tail -5) > $O/r06_v1_t_gemm.log
(timeout 1200 python -m pytest tests/test_decoder_model_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -15) > $O/r06_v1_t_models.log
for rep in 1 2; do
  for v in "--no-multi" "" "--chain"; do
    timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$v' or 'multi', 'ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'gemm_TF', r['achieved'], 'launches', r['launches'], 'check', d.get('step_check'))" >> $O/r06_v1_ab.txt
  done
done
cat $O/r06_v1_t_multi.log $O/r06_v1_multi_bench.log $O/r06_v1_t_gemm.log $O/r06_v1_t_models.log $O/r06_v1_ab.txt
