#!/bin/bash
# round 3, trip R: wait-free fine stamps of the head_dim 64 tile (where do 2273 cycles go when the vector unit is active 1414?)
O=gpurun_out/r3r; mkdir -p $O
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_fs64.so timeout 300 python tools/w64_stamps.py 1 16 8192 64 bf16 fs 0,4,8,12,16,20,24,28 256 > $O/fs64.txt 2>$O/err.txt; cat $O/fs64.txt; tail -3 $O/err.txt
