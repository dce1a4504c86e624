#!/bin/bash
# round 5, trip m: mask pre-passes (prefix scan in the attention kernel, re-pack riding in the cast launch), cost-model routing with the fp16 factor
O=gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_w64_masks.py tests/test_gpu_pv16_range.py tests/test_gpu_routing.py tests/test_gpu_streams.py -q -x 2>&1 | tail -8 | cut -c1-600
R=$GRAFT_REPO_ROOT
for kind in blockdiag window_tensor padding; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$kind -- python3 $R/tools/run_masked.py 100 $kind > $R/$O/out_$kind.txt 2>$R/$O/prof_err.txt )
  echo "== $kind $(tail -1 $O/out_$kind.txt)"; python3 - $O/trace_$kind <<'PY'
import csv,glob,sys
tot=0
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and float(r[4])>0.5 and int(r[1])>=50: print('  ',r[0][:90], r[1], round(float(r[3])/1000,2)); tot+=float(r[3])/1000
print('   sum', round(tot,1))
PY
done
