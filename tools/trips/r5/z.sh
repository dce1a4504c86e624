#!/bin/bash
# trip z: when does the quantiser's V exchange time out?
O=gpurun_out/r5z; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
echo "new only, default"; python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --quant 2 new=$L 2>&1 | grep shape
echo "new only, eager";   python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --quant 2 new=$L 2>&1 | grep shape
echo "r4 + new(wait=100)"; python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --quant 2 r4=tools/lab_bin/libMFAFFI_r4.so "new=$L:cast_wait_us=100" 2>&1 | grep shape
echo "new(wait=1000)"; python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --quant 2 "new=$L:cast_wait_us=1000" 2>&1 | grep shape
echo "new(wait=1000) eager"; python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --quant 2 "new=$L:cast_wait_us=1000" 2>&1 | grep shape
