#!/bin/bash
# round 3, trip D: quantiser (bit-exactness + time), rocprofv3 kernel stats of the headline, PMC passes (SQ, FETCH, WRITE)
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_quantized.py tests/test_gpu_fp8pv.py -q -x > $O/quant_tests.txt 2>&1; tail -3 $O/quant_tests.txt
AB="timeout 300 python tools/ab_inproc.py"
LIBS="base=tools/lab_bin/libMFAFFI_base.so new=intree"
$AB --quant 2 $LIBS > $O/ab_flux_i8.json 2>$O/ab_err.txt
$AB --quant 2 --shape 1,16,8192,128 $LIBS > $O/ab_cfg4_i8.json 2>>$O/ab_err.txt
$AB --quant 3 --shape 1,16,8192,128 $LIBS > $O/ab_cfg4_f8.json 2>>$O/ab_err.txt
cat $O/ab_*.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -- python3 bench.py --steps 20 --warmup 5 --headline-only --no-graph > $O/bench_under_rocprof.json 2>$O/prof_err.txt
tail -c 400 $O/bench_under_rocprof.json
python3 tools/run_pair.py 20 1 24 4096 128 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pair -- python3 tools/run_pair.py 20 1 24 4096 128 > $O/pair.txt 2>>$O/prof_err.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 tools/run_fwd.py 10 > $O/pmc_sq.txt 2>>$O/prof_err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/run_fwd.py 10 > $O/pmc_fetch.txt 2>>$O/prof_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/run_fwd.py 10 > $O/pmc_write.txt 2>>$O/prof_err.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_s8192 -- python3 tools/run_fwd.py 10 1 16 8192 128 > /dev/null 2>>$O/prof_err.txt
python3 tools/pmc_summary.py $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_sq_s8192 > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
find $O/prof_fwd -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fwd_kernel_stats.csv
find $O/prof_pair -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/pair_kernel_stats.csv
head -5 $O/fwd_kernel_stats.csv; head -6 $O/pair_kernel_stats.csv
# keep the merged output small: drop the raw traces
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O
