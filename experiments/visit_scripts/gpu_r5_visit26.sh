#!/bin/bash
# round-5 visit 26: PMC of the three attention kernels on the final tree (persistent dK/dV)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
./tools/pmc_run.sh attn tools/attn_bench.py all; cat gpurun_out/pmc_attn_summary.txt | grep -E "^bridge|GRBM|INSTS_VALU|INSTS_MFMA|INSTS_SALU|INSTS_LDS|MFMA_BUSY|BANK_CONFLICT|IDX_ACTIVE"
python tools/pmc_clock.py gpurun_out/pmc_attn_3 | tee gpurun_out/pmc_attn_clock.txt
rm -rf gpurun_out/pmc_attn_[123]
