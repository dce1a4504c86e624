#!/bin/bash
# round 3, trip A: 4x4x4 row-sum probe, parity of the msum / lazy kernels, same-box A/B against the round-2 kernel, bench
O=gpurun_out/r3a; mkdir -p $O
./tools/lab_bin/msum_probe > $O/probe.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_w64.py -q -x > $O/w64_tests.txt 2>&1
tail -5 $O/w64_tests.txt
AB="timeout 300 python tools/ab_inproc.py"
$AB --parity base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_flux_lazy.json 2>$O/ab_err.txt
UMFA_W64_LAZY=0 $AB --parity base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_flux_msum_tau6.json 2>>$O/ab_err.txt
UMFA_W64_TAU=0 $AB --parity base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_flux_tau0.json 2>>$O/ab_err.txt
$AB --dtype fp16 --parity base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_flux_fp16.json 2>>$O/ab_err.txt
$AB --shape 1,16,8192,128 base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_s8192.json 2>>$O/ab_err.txt
$AB --shape 4,16,8192,128 --causal --rounds 6 --inner 5 base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_causal.json 2>>$O/ab_err.txt
$AB --quant 2 base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_flux_i8.json 2>>$O/ab_err.txt
$AB --quant 2 --shape 1,16,8192,128 base=tools/lab_bin/libMFAFFI_base.so new=intree > $O/ab_cfg4_i8.json 2>>$O/ab_err.txt
cat $O/probe.txt $O/ab_*.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_regime.json 2>$O/bench_err.txt
tail -c 1500 $O/bench_driver_regime.json
