#!/bin/bash
# trip ac: forced part counts of the split-KV plan after the fold's read-ahead (is the plan's "3 tiles per part" still the price?)
O=gpurun_out/r6ac; mkdir -p $O
for sh in "1 32 1 8192 128" "16 8 1 4096 128" "1 8 1 131072 128" "1 32 1 32768 128" "4 8 1 8192 128" "2 16 16 4096 64"; do
  for k in 0 2 4 8 16 32; do
    echo -n "force_split=$k  "; UMFA_FORCE_SPLIT=$k timeout 120 python3 tools/bench_decode.py $sh 2>&1 | tail -1
  done
done | tee $O/decode_forced_parts.txt
