#!/bin/bash
O=gpurun_out/r3ds; mkdir -p $O
timeout 900 python tools/lab/ds_store_probe.py 2>&1 | tail -7
export TMPDIR=/tmp
UMFA_BWD_DS_STORE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/run_bwd.py 1 24 4096 128 20 > /dev/null 2>$O/err.txt
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/ds_kernel_stats.csv; cut -c1-160 $O/ds_kernel_stats.csv | head -8
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
