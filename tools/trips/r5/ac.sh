#!/bin/bash
# trip ac: where does the sliding-window launch (+-512 at FLUX) spend its time?  stamps + kernel trace
O=gpurun_out/r5ac; mkdir -p $O
export TMPDIR=/tmp
L=tools/lab_bin/libMFAFFI_stamps.so
for a in "1 24 4096 128 $L window512" "1 24 4096 128 $L window1024" "1 6 16384 128 $L window512"; do
  echo "== $a" >> $O/wg.txt
  python3 tools/lab/w64_wg_times.py $a 2>&1 | grep -v amdgpu.ids >> $O/wg.txt
done
grep "==\|span\|pro/loop" $O/wg.txt
