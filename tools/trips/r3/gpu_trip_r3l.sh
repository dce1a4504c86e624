#!/bin/bash
# round 3, trip L: what the MFMA SHAPE is worth at the power cap -- timing-only variant (two 16x16x32 per 32x32x16, same FLOPs /
# LDS reads / vector work / operand statistics, results garbage) against the same build without it; + D = 64 gate borders
O=gpurun_out/r3l; mkdir -p $O
for sh in "1,24,4096,128" "1,16,8192,128" "1,4,32768,128"; do
  timeout 600 python tools/ab_inproc.py --shape $sh --rounds 14 --inner 20 ctl=tools/lab_bin/libMFAFFI_ctl.so s16=tools/lab_bin/libMFAFFI_shape16.so >> $O/ab_shape16.jsonl 2>>$O/ab_err.txt
done
timeout 600 python tools/ab_inproc.py --shape 4,16,8192,128 --causal --rounds 10 --inner 10 ctl=tools/lab_bin/libMFAFFI_ctl.so s16=tools/lab_bin/libMFAFFI_shape16.so >> $O/ab_shape16.jsonl 2>>$O/ab_err.txt
cat $O/ab_shape16.jsonl; tail -3 $O/ab_err.txt
timeout 600 python tools/lab/d64_probe.py 1,16,2048,False,bf16 1,8,2048,False,bf16 1,40,1024,False,bf16 2,24,1024,False,bf16 1,24,1024,False,bf16 1,96,512,False,bf16 5,16,1024,True,bf16 2,24,2048,True,bf16 1,16,4096,True,bf16 1,8,8192,True,bf16 > $O/d64_gate.jsonl 2>>$O/ab_err.txt
cat $O/d64_gate.jsonl
