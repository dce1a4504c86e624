#!/bin/bash
# round 3, trip W: lazy-mode UNDERFLOW (rows that start a segment on the reference 0): the old kernel must fail the new test, the new one pass
O=gpurun_out/r3w; mkdir -p $O
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_ctl.so timeout 600 python -m pytest tests/test_gpu_w64.py -q -k "underflow" > $O/old_lib.txt 2>&1; tail -6 $O/old_lib.txt
timeout 900 python -m pytest tests/test_gpu_w64.py -x -q > $O/new_lib.txt 2>&1; tail -6 $O/new_lib.txt
timeout 600 python tools/ab_inproc.py --shape 4,16,8192,128 --causal --rounds 8 --inner 10 ctl=tools/lab_bin/libMFAFFI_ctl.so new=intree > $O/ab_causal.json 2>$O/err.txt; cat $O/ab_causal.json
timeout 600 python tools/ab_inproc.py --shape 1,24,4096,128 --causal --rounds 8 --inner 20 ctl=tools/lab_bin/libMFAFFI_ctl.so new=intree >> $O/ab_causal.json 2>>$O/err.txt; tail -1 $O/ab_causal.json; tail -2 $O/err.txt
