#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk -- python $R/tools/attn_bench.py all > $R/gpurun_out/pmc_clk.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk2 -- python $R/tools/attn_bench.py all > $R/gpurun_out/pmc_clk2.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk3 -- python $R/tools/gemm_one.py 8192 8192 8192 > $R/gpurun_out/pmc_clk3.log 2>&1
cd $R
python tools/pmc_clock.py gpurun_out/pmc_clk | tee gpurun_out/attn_clock.txt
python tools/pmc_clock.py gpurun_out/pmc_clk3 | tee -a gpurun_out/attn_clock.txt
python tools/pmc_summary.py gpurun_out/pmc_clk gpurun_out/pmc_clk2 | tee -a gpurun_out/attn_clock.txt
tail -3 gpurun_out/pmc_clk.log
