#!/bin/bash
# trip aw: what would the lazy tile bodies buy a block-diagonal (all listed tiles open) mask?  timing-only build that forces them
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for s in 1,24,4096,128 4,16,4096,128; do
python3 tools/ab_inproc.py --shape $s --out fp32 --mask blockdiag --graph new=$L lazy=tools/lab_bin/libMFAFFI_masklazy.so 2>&1 | grep shape | cut -c1-330
python3 tools/ab_inproc.py --shape $s --out fp32 --mask window_tensor --graph new=$L lazy=tools/lab_bin/libMFAFFI_masklazy.so 2>&1 | grep shape | cut -c1-330
done
