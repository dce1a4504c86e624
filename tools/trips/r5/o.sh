#!/bin/bash
# round 5, trip o: does a head group's dS stay in the last-level cache between bwd16_dkdv (producer) and bwd16_dq_gemm (consumer)?
# kernel times of the dS-store backward at 3 / 6 / 12 / 24 heads (100 / 201 / 403 / 805 MB of dS), per head, next to the recomputing form
O=gpurun_out/r5o; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for H in 3 6 12 24; do
 for form in ds rec; do
  if [ $form = ds ]; then export UMFA_BWD_DS_STORE=1; else unset UMFA_BWD_DS_STORE; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_${form}_h$H -- python3 $R/tools/run_bwd.py 1 $H 4096 128 40 > /dev/null 2>$R/$O/prof_err.txt )
  echo "== H $H $form"; python3 - $O/trace_${form}_h$H $H <<'PY'
import csv,glob,sys
H=int(sys.argv[2]); tot=0
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and ('bwd16' in r[0]) and int(r[1])>=30:
            us=float(r[3])/1000; tot+=us
            print('  ',r[0][:60].replace('umfa::',''), r[1], round(us,1), 'per head', round(us/H,2))
print('   sum', round(tot,1), 'per head', round(tot/H,2))
PY
 done
done
