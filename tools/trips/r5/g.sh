#!/bin/bash
# round 5, trip g: one-wave-per-block quantiser -- bit-exactness tests, kernel trace of the int8 FLUX call (new form / workgroup form)
O=gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_quantized.py tests/test_gpu_fp8pv.py -q -x 2>&1 | tail -5 | cut -c1-250
R=$GRAFT_REPO_ROOT
for form in wave wg; do
  if [ $form = wg ]; then export UMFA_QUANT_BLOCK_WG=1; else unset UMFA_QUANT_BLOCK_WG; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$form -- python3 $R/tools/run_i8.py 200 blockwise > /dev/null 2>$R/$O/prof_err.txt )
  echo "== $form"; python3 - $O/trace_$form <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if r and r[0]!='Name' and float(r[4])>0.5: print('  ',r[0][:60], r[1], round(float(r[3])/1000,2))
PY
done
unset UMFA_QUANT_BLOCK_WG
python tools/ab_inproc.py --graph --rounds 10 --inner 20 --quant 2 --shape 1,24,4096,128 r4=tools/lab_bin/libMFAFFI_r4.so new=intree | cut -c1-400
python tools/ab_inproc.py --graph --rounds 10 --inner 20 --quant 2 --shape 1,16,8192,128 r4=tools/lab_bin/libMFAFFI_r4.so new=intree | cut -c1-400
