#!/bin/bash
# trip z (fourth session, final build): kernel trace of the bench's headline; kernel trace of the chunked synchronous forward on host buffers
# (tools/lab/host_boundary_probe.py's FLUX call, 8 chunks of 3 heads: the attention + cast kernels per chunk)
O=gpurun_out/r6z; mkdir -p $O
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof.json 2>$R/$O/prof_err.txt )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_host -- python3 $R/tools/run_host_forward.py 20 > $R/$O/host_forward.txt 2>>$R/$O/prof_err.txt )
find $O/trace_host -name "*kernel_stats.csv" -exec cp {} $O/host_forward_kernel_stats.csv \;
rm -rf $O/trace $O/trace_host
head -4 $O/fwd_kernel_stats.csv | cut -c1-200; head -5 $O/host_forward_kernel_stats.csv | cut -c1-200; cat $O/host_forward.txt; tail -2 $O/prof_err.txt
