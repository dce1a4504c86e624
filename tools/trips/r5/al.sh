#!/bin/bash
# trip al: non-temporal output stores?
O=gpurun_out/r5al; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for s in 1,24,4096,128 1,32,4096,128 1,16,8192,128; do
python3 tools/ab_inproc.py --shape $s --out fp32 --graph new=$L nt=tools/lab_bin/libMFAFFI_ntstore.so 2>&1 | grep shape | tee -a $O/nt.txt
done
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out same --graph new=$L nt=tools/lab_bin/libMFAFFI_ntstore.so 2>&1 | grep shape | tee -a $O/nt.txt
