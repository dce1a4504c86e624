#!/bin/bash
# trip x: kernel trace of the quantised backward (what the dO amax + cast cost), block-diagonal mask: bench entry vs tools/ab_inproc.py on one box
O=gpurun_out/r5x; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/lab/qbwd_stream_loop.py 10 > $R/$O/out.txt 2>$R/$O/prof_err.txt )
python3 - $O/trace <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and float(r[4])>0.2: print('  ',r[0][:110], r[1], round(float(r[3])/1000,1))
PY
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --mask blockdiag --graph new=universal-metal-flash-attention_amd/lib/libMFAFFI.so > $O/ab_blockdiag.txt 2>&1; tail -1 $O/ab_blockdiag.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5x/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('cfg4_int8_bwd','cfg3_flux_bf16_mask_blockdiag','cfg3_flux_bf16_mask_padding'):
    print(k, d['configs'].get(k))
PY
