#!/bin/bash
# trip v: ablations of the 128-row kernel at config 2 under the PAIRED schedule (every CU holds two 9-tile workgroups all the way: what do two co-resident workgroups wait for?)
O=gpurun_out/r6v; mkdir -p $O
for n in base pv0 pv2 noload noqk nopv noexp nobar nomma noarith noloadbar; do echo "== $n"; timeout 60 tools/lab_bin/abl_$n 16 1024 30 4 0 | grep -v "^ *[0-9]" | grep -v "^block"; done 2>&1 | tee $O/ablations_paired.txt
