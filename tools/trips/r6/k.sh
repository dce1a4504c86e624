#!/bin/bash
# trip k: kernel trace + SQ counters of the bf16 backward at the FLUX shape (where do its 0.6 ms go?)
O=gpurun_out/r6k; mkdir -p $O
R=$PWD
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/run_bwd.py 1 24 4096 128 30 > /dev/null 2>$R/$O/err.txt )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/bwd_kernel_stats.csv \;
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $SQ --output-format csv -d $R/$O/pmc_sq -- python3 $R/tools/run_bwd.py 1 24 4096 128 10 > /dev/null 2>>$R/$O/err.txt )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/$O/pmc_lds -- python3 $R/tools/run_bwd.py 1 24 4096 128 10 > /dev/null 2>>$R/$O/err.txt )
python3 tools/pmc_summary.py $O/pmc_sq $O/pmc_lds | tee $O/pmc_summary.txt
head -8 $O/bwd_kernel_stats.csv | cut -c1-160
rm -rf $O/trace $O/pmc_sq $O/pmc_lds
