#!/bin/bash
# PMC passes over the attention microbench (counters only + kernel trace, as the pool requires)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-all}
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -- python $R/tools/attn_bench.py $W > $R/gpurun_out/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -- python $R/tools/attn_bench.py $W > $R/gpurun_out/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc3 -- python $R/tools/attn_bench.py $W > $R/gpurun_out/pmc3.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 | tee gpurun_out/pmc_summary.txt
