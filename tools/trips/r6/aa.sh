#!/bin/bash
# trip aa: decode-shaped launches (few query rows, long K / V): achieved HBM rate against the 6.29 TB/s copy rate
O=gpurun_out/r6aa; mkdir -p $O
for sh in "8 32 1 8192 128" "1 32 1 8192 128" "1 32 1 32768 128" "32 32 1 2048 128" "8 32 1 8192 64" "4 32 8 8192 128" "16 8 1 4096 128" "1 8 1 131072 128" "64 8 1 1024 128"; do
  timeout 120 python3 tools/bench_decode.py $sh 2>&1 | tail -1
done | tee $O/decode.txt
