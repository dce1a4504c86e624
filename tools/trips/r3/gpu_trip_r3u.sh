#!/bin/bash
O=gpurun_out/r3u; mkdir -p $O
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_fs64b.so timeout 300 python tools/w64_stamps.py 1 16 8192 64 bf16 fs 0,1,2,3,28,29,30,31 256 > $O/fs64b.txt 2>$O/err.txt; cat $O/fs64b.txt
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_fs64b.so timeout 300 python tools/w64_stamps.py 1 16 8192 128 bf16 fs 0,1,2,3,28,29,30,31 256 > $O/fs128b.txt 2>>$O/err.txt; cat $O/fs128b.txt; tail -2 $O/err.txt
