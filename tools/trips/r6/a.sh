#!/bin/bash
# trip a (round 6): the hygiene batch on hardware -- the touched tests, the full bench line, and the counter passes the round-5 review asked to refresh
# on the SHIPPED default kernel (fa_fwd16_w64_bf16pv16): SQ group, FETCH_SIZE / WRITE_SIZE (separate passes), kernel trace; eager call overhead
O=gpurun_out/r6a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python3 -m pytest tests/test_gpu_quantized.py tests/test_gpu_pv16_range.py tests/test_gpu_configs.py tests/test_gpu_backward.py tests/test_gpu_w64_masks.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o fwd -- python3 bench.py --steps 20 --warmup 5 --headline-only --no-graph > $O/bench_under_rocprof.json 2>/dev/null
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 tools/run_fwd.py 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_s8192 -- python3 tools/run_fwd.py 10 1 16 8192 128 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/run_fwd.py 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/run_fwd.py 10 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pmc_sq $O/pmc_sq_s8192 $O/pmc_fetch $O/pmc_write | tee $O/pmc_summary.txt
rm -rf $O/trace $O/pmc_sq $O/pmc_sq_s8192 $O/pmc_fetch $O/pmc_write
python3 tools/lab/eager_overhead.py 2>&1 | grep "per call" | tee $O/eager_overhead.txt
