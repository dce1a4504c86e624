#!/bin/bash
# trip e: counter passes on the SHIPPED default kernel (fa_fwd16_w64_bf16pv16) -- SQ group (FLUX and B1 H16 S8192), FETCH_SIZE / WRITE_SIZE in separate
# passes, kernel trace of the bench's headline; and the same SQ group on the additive-mask kernel (tools/run_masked.py bias)
O=gpurun_out/r6e; mkdir -p $O
R=$PWD
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof.json 2>$R/$O/prof_err.txt )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $SQ --output-format csv -d $R/$O/pmc_sq -- python3 $R/tools/run_fwd.py 10 > /dev/null 2>>$R/$O/prof_err.txt )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $SQ --output-format csv -d $R/$O/pmc_sq_s8192 -- python3 $R/tools/run_fwd.py 10 1 16 8192 128 > /dev/null 2>>$R/$O/prof_err.txt )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/tools/run_fwd.py 10 > /dev/null 2>>$R/$O/prof_err.txt )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/tools/run_fwd.py 10 > /dev/null 2>>$R/$O/prof_err.txt )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $SQ --output-format csv -d $R/$O/pmc_sq_bias -- python3 $R/tools/run_bias.py 10 > /dev/null 2>>$R/$O/prof_err.txt )
python3 tools/pmc_summary.py $O/pmc_sq $O/pmc_sq_s8192 $O/pmc_fetch $O/pmc_write $O/pmc_sq_bias | tee $O/pmc_summary.txt
rm -rf $O/trace $O/pmc_sq $O/pmc_sq_s8192 $O/pmc_fetch $O/pmc_write $O/pmc_sq_bias
tail -3 $O/prof_err.txt
