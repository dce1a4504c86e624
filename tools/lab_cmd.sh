cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_w64.py -m gpu -x -q 2>&1 | tail -3
for s in "1 24 4096 128" "1 8 4096 128" "1 16 8192 128" "2 24 4096 128"; do python tools/bench_one.py $s; done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r1_w64b_pmc_fetch -- python3 tools/run_fwd.py 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r1_w64b_pmc_write -- python3 tools/run_fwd.py 5 > /dev/null 2>&1
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
