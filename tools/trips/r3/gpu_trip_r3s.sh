#!/bin/bash
# round 3, trip S: tile-end wait (vmcnt: LDS-DMA landing) vs barrier (waiting for the other waves) per workgroup, head_dim 64 and 128
O=gpurun_out/r3s; mkdir -p $O
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_ws64.so timeout 300 python tools/w64_stamps.py 1 16 8192 64 > $O/ws64.txt 2>$O/err.txt; cat $O/ws64.txt
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_ws64.so timeout 300 python tools/w64_stamps.py 1 16 8192 128 > $O/ws128.txt 2>>$O/err.txt; cat $O/ws128.txt; tail -3 $O/err.txt
