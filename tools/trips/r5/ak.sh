#!/bin/bash
# trip ak: what do the O stores cost?  (timing-only build without them) FLUX, window +-512, H12, and fp32 vs bf16 O on the real build
O=gpurun_out/r5ak; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for s in 1,24,4096,128 1,12,4096,128 1,32,4096,128; do
python3 tools/ab_inproc.py --shape $s --out fp32 --graph new=$L nostore=tools/lab_bin/libMFAFFI_nostore.so 2>&1 | grep shape | tee -a $O/nostore.txt
python3 tools/ab_inproc.py --shape $s --out same --graph new=$L nostore=tools/lab_bin/libMFAFFI_nostore.so 2>&1 | grep shape | tee -a $O/nostore.txt
done
