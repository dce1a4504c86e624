#!/bin/bash
# round-2 first probe: parity numbers at full sizes, TAU A/B, few-head shard shapes, baseline bench
mkdir -p gpurun_out/r2a
cd /root/repo
nproc > gpurun_out/r2a/nproc.txt
python tools/parity_probe.py > gpurun_out/r2a/parity.json 2> gpurun_out/r2a/parity.log
python tools/ab_inproc.py --parity --out fp32 tau6=intree tau0=tools/lab_bin/libMFAFFI_tau0.so > gpurun_out/r2a/tau_flux.json 2> gpurun_out/r2a/tau_flux.err
python tools/ab_inproc.py --parity --out fp32 --shape 1,16,8192,128 tau6=intree tau0=tools/lab_bin/libMFAFFI_tau0.so > gpurun_out/r2a/tau_s8192.json 2>> gpurun_out/r2a/tau_flux.err
for sh in "1 3 4096 128" "1 6 4096 128" "1 12 4096 128" "1 24 4096 128" "1 4 32768 128" "1 32 32768 128" "4 16 1024 64 causal"; do
  python tools/bench_one.py $sh >> gpurun_out/r2a/shapes.txt 2>&1
  UMFA_NO_W64=1 python tools/bench_one.py $sh >> gpurun_out/r2a/shapes_no_w64.txt 2>&1
done
python bench.py --steps 50 --warmup 10 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -3 gpurun_out/r2a/parity.log; cat gpurun_out/r2a/tau_flux.json gpurun_out/r2a/tau_s8192.json gpurun_out/r2a/shapes.txt gpurun_out/r2a/shapes_no_w64.txt
