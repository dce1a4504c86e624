#!/bin/bash
# round 4, trip W: dS-store backward in head groups (producer / consumer alternate so dS may stay in the last-level cache): wall time per backward
O=gpurun_out/r4w; mkdir -p $O
export TMPDIR=/tmp
for g in 0 12 8 6 4 2; do
  UMFA_LAB_DS_GROUP=$g timeout 300 python tools/lab/ds_store_probe.py 2>/dev/null | tail -1 > $O/probe_group$g.txt; echo "group $g: $(cat $O/probe_group$g.txt)"
done
R=$GRAFT_REPO_ROOT
( cd /tmp && UMFA_BWD_DS_STORE=1 UMFA_LAB_DS_GROUP=8 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/run_bwd.py 1 24 4096 128 20 > $R/$O/run.txt 2>&1 )
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats_group8.csv; grep bwd16 $O/stats_group8.csv | cut -d, -f1-4 | cut -c1-150; rm -rf $O/trace; find $O -name "*.db" -delete
