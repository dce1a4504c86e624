#!/bin/bash
# round 4, trip B: the default bf16 forward = fp16 P V with a fast V cast pre-pass (w64 kernels) / in-kernel conversion (128-row
# kernel): the whole GPU suite, the A/B probe against the bf16 P V kernels, the bench line, kernel trace of the bench
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 900 python tools/lab/pv16_probe.py > $O/pv16_probe.jsonl 2>$O/probe_err.txt; cat $O/pv16_probe.jsonl; tail -3 $O/probe_err.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; tail -c 3000 $O/bench.json; tail -3 $O/bench_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --headline-only --no-graph > $O/bench_under_rocprof.json 2>$O/prof_err.txt
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fwd_kernel_stats.csv; cut -c1-200 $O/fwd_kernel_stats.csv | head -6
rm -rf $O/trace; find $O -name "*.db" -delete
