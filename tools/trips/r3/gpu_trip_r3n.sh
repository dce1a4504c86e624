#!/bin/bash
# round 3, trip N: PMC of the head_dim 64 kernel (what bounds a 32-MFMA tile?)
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_d64 -- python3 tools/run_fwd.py 10 1 16 8192 64 > /dev/null 2>>$O/prof_err.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq2_d64 -- python3 tools/run_fwd.py 10 1 16 8192 64 > /dev/null 2>>$O/prof_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_clk_d64 -- python3 tools/run_fwd.py 10 1 16 8192 64 > /dev/null 2>>$O/prof_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_clk_d128 -- python3 tools/run_fwd.py 10 1 16 8192 128 > /dev/null 2>>$O/prof_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d64 -- python3 tools/run_fwd.py 20 1 16 8192 64 > /dev/null 2>>$O/prof_err.txt
python3 tools/pmc_summary.py $O/pmc_sq_d64 $O/pmc_sq2_d64 $O/pmc_clk_d64 $O/pmc_clk_d128 > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
find $O/prof_d64 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/d64_kernel_stats.csv; head -4 $O/d64_kernel_stats.csv
tail -3 $O/prof_err.txt
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
