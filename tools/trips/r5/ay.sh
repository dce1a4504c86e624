#!/bin/bash
# trip ay: head_dim 64: the quantised forward (fa_fwd_i8<64>: no one-wave-per-SIMD int8 kernel at 64) against the bf16 forward
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for s in 2,16,4096,64 4,16,1024,64 1,24,8192,64; do
python3 tools/ab_inproc.py --shape $s --out fp32 --graph new=$L 2>&1 | grep shape | cut -c1-250
python3 tools/ab_inproc.py --shape $s --out fp32 --graph --quant 2 new=$L 2>&1 | grep shape | cut -c1-250
done
