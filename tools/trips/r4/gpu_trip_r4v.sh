#!/bin/bash
# round 4, trip V: what the dS stores cost bwd16_dkdv: kernel stats with (a) the stores as built, (b) every store into tile 0 (no HBM write
# stream), (c) non-temporal stores; and the recomputing form on the same box
O=gpurun_out/r4v; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, env...
  name=$1; shift
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$name -- python3 $R/tools/run_bwd.py 1 24 4096 128 20 > $R/$O/run_$name.txt 2>&1 )
  find $O/trace_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats_$name.csv
  echo "== $name"; grep "bwd16" $O/stats_$name.csv | cut -d, -f1-4 | cut -c1-150
  rm -rf $O/trace_$name
}
run recompute UMFA_BWD_DS_STORE=0
run ds_store UMFA_BWD_DS_STORE=1
run ds_store_tile0 UMFA_BWD_DS_STORE=1 UMFA_LAB_DS=1
run ds_store_nt UMFA_BWD_DS_STORE=1 UMFA_LIBRARY=$R/tools/lab_bin/libMFAFFI_dsnt.so
find $O -name "*.db" -delete
