#!/bin/bash
# trip ae: decode-shaped launches with a deeper LDS-DMA ring (lab: -DUMFA_LAB_NS=3 / 4 on the non-causal 128-row kernels) + the new fold constant
O=gpurun_out/r6ae; mkdir -p $O
for sh in "8 32 1 8192 128" "1 32 1 8192 128" "32 32 1 2048 128" "8 32 1 8192 64" "4 32 8 8192 128" "1 8 1 131072 128" "64 8 1 1024 128" "16 32 1 4096 128"; do
  for lib in "" tools/lab_bin/libMFAFFI_ns3.so tools/lab_bin/libMFAFFI_ns4.so; do
    echo -n "lib=${lib:-product}  "; UMFA_LIBRARY=$lib timeout 120 python3 tools/bench_decode.py $sh 2>&1 | tail -1
  done
done | tee $O/decode_ring_depth.txt
