#!/bin/bash
# trip aj: the strong-scaling shards of the FLUX problem on one GPU (heads 24 / 12 / 6 / 3 = N 1 / 2 / 4 / 8)
O=gpurun_out/r5aj; mkdir -p $O
for h in 24 12 6 3; do
python3 tools/ab_inproc.py --shape 1,$h,4096,128 --out fp32 --graph new=universal-metal-flash-attention_amd/lib/libMFAFFI.so 2>&1 | grep shape | tee -a $O/shards.txt
done
