#!/bin/bash
O=gpurun_out/r5t; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/lab/i8_mask_probe.py 16 8192 > $R/$O/out.txt 2>$R/$O/prof_err.txt )
python3 - $O/trace <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and float(r[4])>0.3: print('  ',r[0][:100], r[1], round(float(r[3])/1000,1))
PY
