#!/bin/bash
# trip as: the 128-row bf16 kernel at head_dim 128 non-causal: 32-key tiles (default) against 64-key tiles (option bn64), on shapes the dispatcher gives it
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
cp $L /tmp/lib2.so
for s in 2,16,1000,128 8,8,200,128 4,8,512,128 1,24,4096,128 16,16,256,128; do
python3 tools/ab_inproc.py --shape $s --out fp32 --graph "bn32=$L:no_w64=1" "bn64=/tmp/lib2.so:no_w64=1,bn64=1" 2>&1 | grep shape | cut -c1-330
done
