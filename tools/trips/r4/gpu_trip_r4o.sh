#!/bin/bash
# round 4, trip O2: where the cycles of a tile iteration of the head_dim-64 kernel go (wave 0 of every workgroup, s_memtime buckets)
O=gpurun_out/r4o; mkdir -p $O
for v in loop_ks1 loop_ks1_fine loop_ks2 loop_ks2_fine; do timeout 60 tools/lab_bin/cfg2_$v 16 1024 50 4 > $O/$v.txt 2>&1; echo $v; tail -4 $O/$v.txt | cut -c1-400; done
