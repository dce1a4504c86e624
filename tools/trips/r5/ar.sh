#!/bin/bash
# trip ar: fa_fwd_i8 unmasked at head_dim 128: 32-key tiles (three workgroups per CU) against 64-key tiles (two)
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for s in 1,16,8192,128 1,24,4096,128 8,8,200,128 2,16,1000,128; do
python3 tools/ab_inproc.py --shape $s --out fp32 --graph --quant 2 "bn32=$L:no_w64=1" "bn64=tools/lab_bin/libMFAFFI_i8bn64.so:no_w64=1" 2>&1 | grep shape | cut -c1-330
done
