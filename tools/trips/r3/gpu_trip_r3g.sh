#!/bin/bash
# round 3, trip G: backward kernel stats + PMC at HEAD
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
python3 tools/run_bwd.py 1 24 4096 128 10 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bwd -- python3 tools/run_bwd.py 1 24 4096 128 10 > $O/run_bwd.txt 2>$O/prof_err.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_bwd -- python3 tools/run_bwd.py 1 24 4096 128 6 > /dev/null 2>>$O/prof_err.txt
find $O/prof_bwd -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bwd16_flux_kernel_stats.csv
python3 tools/pmc_summary.py $O/pmc_bwd > $O/pmc_bwd_summary.txt 2>&1
cut -c1-200 $O/bwd16_flux_kernel_stats.csv | head -6; cat $O/pmc_bwd_summary.txt
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*.db" -delete
