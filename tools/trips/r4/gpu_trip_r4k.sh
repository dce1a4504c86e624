#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_stamps.so UMFA_STRIP_TRUE_MASKS=0 timeout 600 python tools/lab/mask_stamps.py > $O/mask_stamps.txt 2>&1; cat $O/mask_stamps.txt
