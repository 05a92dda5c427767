#!/bin/bash
# PMC passes (counters + kernel trace only) over an arbitrary python command: tools/pmc_run.sh <tag> <python args...>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift; SCRIPT=$1; shift
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_1 -- python $R/$SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}_1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_2 -- python $R/$SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}_2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_3 -- python $R/$SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}_3.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3 > gpurun_out/pmc_${TAG}_summary.txt
