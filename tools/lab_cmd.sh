cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 50 --warmup 10 > gpurun_out/bench_r1_final.json 2> gpurun_out/bench_r1_final.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1_end_stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph > gpurun_out/bench_r1_final_prof.json 2> gpurun_out/bench_r1_final_prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r1_end_pmc_fetch -- python3 tools/run_fwd.py 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r1_end_pmc_write -- python3 tools/run_fwd.py 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/r1_end_pmc_sq -- python3 tools/run_fwd.py 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r1_end_pmc_lds -- python3 tools/run_fwd.py 5 > /dev/null 2>&1
cat gpurun_out/bench_r1_final.json | cut -c1-1600
