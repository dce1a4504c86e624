#!/bin/bash
# trip n: config 2 -- the key-split form (KS = 2, lab: -DUMFA_D64_FORMS, option ksplit) with the two key halves of a row-wave on ADJACENT waves (different
# SIMDs) instead of waves rw and rw + 4 (the same SIMD): round 4 measured the form null and noted "its two halves queue for one SIMD"
O=gpurun_out/r6n; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for sh in 4,16,1024,64 8,16,512,64 2,16,2048,64 4,16,2048,64; do
python3 tools/ab_inproc.py --shape $sh --causal --graph --inner 100 --rounds 10 --parity base=$L "ks_old=tools/lab_bin/libMFAFFI_ks_old.so:ksplit=1" "ks_adj=tools/lab_bin/libMFAFFI_ks_adj.so:ksplit=1" "pipe=tools/lab_bin/libMFAFFI_ks_old.so" 2>&1 | grep shape | tee -a $O/ab_ksplit_adjacent.jsonl
done
python3 tools/ab_inproc.py --shape 8,16,1024,64 --graph --inner 50 --rounds 8 base=$L "ks_old=tools/lab_bin/libMFAFFI_ks_old.so:ksplit=1" "ks_adj=tools/lab_bin/libMFAFFI_ks_adj.so:ksplit=1" 2>&1 | grep shape | tee -a $O/ab_ksplit_adjacent.jsonl
