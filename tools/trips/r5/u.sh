#!/bin/bash
# trip u: where does the block-diagonal masked launch spend its time?  per-workgroup stamps (stamps build) at FLUX and at 4x the length
O=gpurun_out/r5u; mkdir -p $O
export TMPDIR=/tmp
L=tools/lab_bin/libMFAFFI_stamps.so
for a in "1 24 4096 128 $L blockdiag4" "1 6 16384 128 $L blockdiag4" "1 24 4096 128 $L blockdiag1" "1 24 4096 128 $L"; do
  echo "== $a" >> $O/wg.txt
  python3 tools/lab/w64_wg_times.py $a >> $O/wg.txt 2>&1
done
python3 tools/ab_inproc.py --shape 1,24,4096,128 --mask blockdiag --graph new=universal-metal-flash-attention_amd/lib/libMFAFFI.so > $O/ab_blockdiag.txt 2>&1
tail -40 $O/wg.txt; tail -3 $O/ab_blockdiag.txt
