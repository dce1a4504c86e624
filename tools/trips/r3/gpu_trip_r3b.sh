#!/bin/bash
# round 3, trip B: lazy with / without matrix-pipe row sums vs the round-2 kernel (time + fp32-O parity), settle probe
O=gpurun_out/r3b; mkdir -p $O
LIBS="base=tools/lab_bin/libMFAFFI_base.so msum=intree nomsum=tools/lab_bin/libMFAFFI_nomsum.so"
AB="timeout 300 python tools/ab_inproc.py"
$AB --out fp32 --parity $LIBS > $O/ab_flux_lazy.json 2>$O/ab_err.txt
UMFA_W64_LAZY=0 $AB --out fp32 --parity $LIBS > $O/ab_flux_tau6.json 2>>$O/ab_err.txt
UMFA_W64_TAU=0 $AB --out fp32 --parity $LIBS > $O/ab_flux_tau0.json 2>>$O/ab_err.txt
UMFA_FORCE_W64=1 $AB --shape 1,64,1024,128 --out fp32 --parity $LIBS > $O/ab_tail_lazy.json 2>>$O/ab_err.txt
UMFA_FORCE_W64=1 UMFA_W64_TAU=0 $AB --shape 1,64,1024,128 --out fp32 --parity $LIBS > $O/ab_tail_tau0.json 2>>$O/ab_err.txt
$AB --shape 1,4,32768,128 --rounds 6 --inner 4 $LIBS > $O/ab_cfg5shard.json 2>>$O/ab_err.txt
$AB --shape 1,16,8192,128 $LIBS > $O/ab_s8192.json 2>>$O/ab_err.txt
$AB --shape 4,16,8192,128 --causal --rounds 6 --inner 5 $LIBS > $O/ab_causal.json 2>>$O/ab_err.txt
$AB --causal $LIBS > $O/ab_flux_causal.json 2>>$O/ab_err.txt
cat $O/ab_*.json
true

