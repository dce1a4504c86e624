#!/bin/bash
# trip av: kernel trace of a training step through the torch SDPA surface
O=gpurun_out/r5av; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/lab/train_step_loop.py 10 > $R/$O/out.txt 2>$R/$O/prof_err.txt )
python3 - $O/trace <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and float(r[4])>0.1: print('  ',r[0][:120], r[1], round(float(r[3])/1000,1))
PY
