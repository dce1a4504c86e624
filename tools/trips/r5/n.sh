#!/bin/bash
# round 5, trip n: masked FLUX calls, round-4 library against this build in one process
O=gpurun_out/r5n; mkdir -p $O
for mk in blockdiag padding window_tensor random; do
python tools/ab_inproc.py --graph --rounds 12 --inner 20 --out fp32 --mask $mk --shape 1,24,4096,128 r4=tools/lab_bin/libMFAFFI_r4.so new=intree | tee -a $O/ab_masks.jsonl | cut -c1-500
done
python tools/ab_inproc.py --graph --rounds 12 --inner 20 --out fp32 --mask padding --shape 4,16,4096,128 r4=tools/lab_bin/libMFAFFI_r4.so new=intree | tee -a $O/ab_masks.jsonl | cut -c1-500
python tools/ab_inproc.py --graph --rounds 12 --inner 20 --out fp32 --mask blockdiag --shape 4,16,4096,128 r4=tools/lab_bin/libMFAFFI_r4.so new=intree | tee -a $O/ab_masks.jsonl | cut -c1-500
