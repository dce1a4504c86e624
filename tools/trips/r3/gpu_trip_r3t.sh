#!/bin/bash
# round 3, trip T: per-tile barrier in the middle of the PV phase (next tile's K fragments read under the remaining MFMAs):
# null at head_dim 128 in round 1 -- at head_dim 64 the tile boundary is 22-29 % of the tile
O=gpurun_out/r3t; mkdir -p $O
for sh in "2,16,4096,64" "1,16,8192,64" "1,24,4096,128" "1,16,8192,128"; do
  timeout 600 python tools/ab_inproc.py --shape $sh --rounds 12 --inner 20 --parity ctl=tools/lab_bin/libMFAFFI_ctl.so midbar=tools/lab_bin/libMFAFFI_midbar.so >> $O/ab_midbar.jsonl 2>>$O/err.txt
done
timeout 600 python tools/ab_inproc.py --shape 4,16,4096,64 --causal --rounds 10 --inner 10 --parity ctl=tools/lab_bin/libMFAFFI_ctl.so midbar=tools/lab_bin/libMFAFFI_midbar.so >> $O/ab_midbar.jsonl 2>>$O/err.txt
cat $O/ab_midbar.jsonl; tail -3 $O/err.txt
