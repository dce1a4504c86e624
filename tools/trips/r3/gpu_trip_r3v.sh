#!/bin/bash
# round 3, trip V: what is in the head_dim 64 tile boundary?  timing-only ablations (results wrong): no barrier / wait, no K-fragment
# reads behind the barrier, no lazy row-sum check, all three
O=gpurun_out/r3v; mkdir -p $O
L="ctl=tools/lab_bin/libMFAFFI_ctl.so nobar=tools/lab_bin/libMFAFFI_nobar.so nokf=tools/lab_bin/libMFAFFI_nokf.so nochk=tools/lab_bin/libMFAFFI_nochk.so all3=tools/lab_bin/libMFAFFI_all3.so"
for sh in "1,16,8192,64" "1,16,8192,128"; do
  timeout 600 python tools/ab_inproc.py --shape $sh --rounds 10 --inner 20 $L >> $O/ab_boundary.jsonl 2>>$O/err.txt
done
cat $O/ab_boundary.jsonl; tail -2 $O/err.txt
