#!/bin/bash
# round 5, trip f: per-workgroup end times of the FLUX launch (stamps build) -- where the tail comes from
O=gpurun_out/r5f; mkdir -p $O
for sh in "1 24 4096 128" "1 16 4096 128" "1 32 4096 128" "1 8 8192 128"; do
  python tools/lab/w64_wg_times.py $sh tools/lab_bin/libMFAFFI_stamps.so 2>/dev/null | tee -a $O/wg_times.txt
done
