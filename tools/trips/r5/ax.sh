#!/bin/bash
# trip ax: the mask kernel on all-true masks (every tile open) against the unmasked kernel: what does the list-driven sweep cost per tile?
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for s in 1,32,4096,128 1,24,4096,128; do
python3 tools/ab_inproc.py --shape $s --out fp32 --graph new=$L 2>&1 | grep shape | cut -c1-250
python3 tools/ab_inproc.py --shape $s --out fp32 --mask alltrue_keys --graph new=$L 2>&1 | grep shape | cut -c1-250
python3 tools/ab_inproc.py --shape $s --out fp32 --mask alltrue_2d --graph new=$L 2>&1 | grep shape | cut -c1-250
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r5ax; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/ab_inproc.py --shape 1,32,4096,128 --out fp32 --mask alltrue_keys --rounds 3 new=$R/$L > /dev/null 2>&1 )
python3 - $O/trace <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and float(r[4])>0.3: print('  ',r[0][:110], r[1], round(float(r[3])/1000,1))
PY
