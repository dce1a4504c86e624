#!/bin/bash
# round 4, trip Q: config 2 with 128-key tiles / with V already fp16 (stamps builds, timing only)
O=gpurun_out/r4q; mkdir -p $O
for v in lab_ks1_ns2 bn128 pv2 bn128_pv2; do timeout 120 tools/lab_bin/cfg2_$v 16 1024 50 4 > $O/stamps_$v.txt 2>&1; echo $v; tail -3 $O/stamps_$v.txt | cut -c1-300; done
