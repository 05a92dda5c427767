#!/bin/bash
# round-5 visit 5: dK/dV switches A/B, CU-budget probe with the per-XCD mask, PMC of the attention kernels and of the dominant GEMM,
# the headline step with the round-4 attention kernels vs this tree on one box
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
out=gpurun_out/v5_dkv_ab.txt; : > $out
keep=$(mktemp); cp libra_amd/lib/liblibra_hip.so $keep
for rep in 1 2; do
  for v in g1nf6 g0nf6 g0nf4 g0nf8; do
    cp ab/libs/$v.so libra_amd/lib/liblibra_hip.so
    echo -n "$v " >> $out; timeout 90 python tools/attn_bench.py bwd 2>&1 | tail -1 >> $out
  done
done
cp $keep libra_amd/lib/liblibra_hip.so
cat $out
timeout 200 python tools/cu_budget_probe.py 2>/dev/null > gpurun_out/cu_budget_probe.txt; cat gpurun_out/cu_budget_probe.txt
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
./tools/pmc_run.sh attn tools/attn_bench.py all; cat gpurun_out/pmc_attn_summary.txt | head -90
python tools/pmc_clock.py gpurun_out/pmc_attn_3 | tee gpurun_out/pmc_attn_clock.txt
./tools/pmc_run.sh gemm tools/gemm_one.py 11760 22016 4096 0 0 20; cat gpurun_out/pmc_gemm_summary.txt | head -40
python tools/pmc_clock.py gpurun_out/pmc_gemm_3 | tee gpurun_out/pmc_gemm_clock.txt
rm -rf gpurun_out/pmc_attn_[123] gpurun_out/pmc_gemm_[123]
./tools/gpu_lib_ab.sh 2 r4attn cur
