#!/bin/bash
# round 3, trip E: quantised backward on the MFMA engine (tests + time), int8 A/B after the bias-tile fix, full suite
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_quantized.py -q -x > $O/bwd_tests.txt 2>&1; tail -12 $O/bwd_tests.txt
AB="timeout 300 python tools/ab_inproc.py"
LIBS="base=tools/lab_bin/libMFAFFI_base.so new=intree"
$AB --quant 2 $LIBS > $O/ab_flux_i8.json 2>$O/ab_err.txt
$AB --quant 2 --shape 1,16,8192,128 $LIBS > $O/ab_cfg4_i8.json 2>>$O/ab_err.txt
cat $O/ab_*.json
timeout 600 python tools/bench_qbwd.py > $O/qbwd.json 2>$O/qbwd_err.txt; cat $O/qbwd.json; tail -3 $O/qbwd_err.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
