#!/bin/bash
# round 5, trip j: where the masked / windowed FLUX calls spend their time (kernel trace per kind)
O=gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in blockdiag window_tensor window padding additive_blockdiag; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$kind -- python3 $R/tools/run_masked.py 100 $kind > $R/$O/out_$kind.txt 2>$R/$O/prof_err.txt )
  echo "== $kind $(tail -1 $O/out_$kind.txt)"; python3 - $O/trace_$kind <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and float(r[4])>0.5 and int(r[1])>=50: print('  ',r[0][:90], r[1], round(float(r[3])/1000,2))
PY
done
