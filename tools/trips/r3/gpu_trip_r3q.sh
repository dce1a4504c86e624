#!/bin/bash
# round 3, trip Q: full GPU suite + the driver's bench command at HEAD + rocprof kernel stats of the same command
O=gpurun_out/r3q; mkdir -p $O
UMFA_PARITY_RECORD=$PWD/$O/parity_record.jsonl timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2>$O/bench_err.txt; tail -c 3000 $O/bench_default.json; tail -3 $O/bench_err.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -- python3 bench.py --steps 20 --warmup 5 --headline-only --no-graph > $O/bench_under_rocprof.json 2>$O/prof_err.txt
find $O/prof_fwd -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fwd_kernel_stats.csv; head -3 $O/fwd_kernel_stats.csv | cut -c1-300
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*.db" -delete
