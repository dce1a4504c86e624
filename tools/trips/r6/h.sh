#!/bin/bash
# trip h: row sums on the matrix pipe (generator W64_MSUM, -DW64_MSUM_ON) at head_dim 64 -- measured slower at head_dim 128 in round 3, where the matrix pipe
# is the busy one; at head_dim 64 the vector unit is (62 % active against 45 %, profiles/r3/pmc_d64.md)
O=gpurun_out/r6h; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for sh in 2,16,4096,64 1,16,8192,64 4,16,2048,64; do
python3 tools/ab_inproc.py --shape $sh --graph --rounds 10 base=$L msum=tools/lab_bin/libMFAFFI_msum.so 2>&1 | grep shape | tee -a $O/ab_msum_d64.jsonl
python3 tools/ab_inproc.py --shape $sh --graph --causal --rounds 10 base=$L msum=tools/lab_bin/libMFAFFI_msum.so 2>&1 | grep shape | tee -a $O/ab_msum_d64.jsonl
done
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --rounds 10 base=$L msum=tools/lab_bin/libMFAFFI_msum.so 2>&1 | grep shape | tee -a $O/ab_msum_d64.jsonl
python3 tools/ab_inproc.py --shape 2,16,4096,64 --parity base=$L msum=tools/lab_bin/libMFAFFI_msum.so 2>&1 | grep shape | tee -a $O/ab_msum_d64.jsonl
