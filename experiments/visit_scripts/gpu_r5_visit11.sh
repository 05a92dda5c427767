#!/bin/bash
# round-5 visit 11: rank-8 bridge weight-gradient pass (fp32 coefficients in LDS + packed FMA) parity and A/B; bench with a CU reserve
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_decoder_kernels_gpu.py -m gpu -q -k "rank_outer" -p no:cacheprovider 2>&1 | tail -2
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2; do for v in ro_old ro_new; do cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so; echo "$v"; timeout 60 python tools/rank_outer_bench.py 2>&1 | grep ncoef; done; done | tee gpurun_out/v11_rank_outer_ab.txt
cp $keep libra_amd/lib/liblibra_hip.so
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --cu-reserve 32 2>gpurun_out/v11_bench_cu.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cu-reserve 32:', d['ms_per_step'], d.get('extra',{}).get('cu_budget'))"
tail -3 gpurun_out/v11_bench_cu.err
