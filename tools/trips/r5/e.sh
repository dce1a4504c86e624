#!/bin/bash
# round 5, trip e: same-box A/B of the round-4 library against this build (config 2, FLUX, decode-like), and where FLUX's efficiency goes (whole rounds vs cut items)
O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
AB="python tools/ab_inproc.py --graph --rounds 16 --inner 50 r4=tools/lab_bin/libMFAFFI_r4.so new=intree"
$AB --shape 4,16,1024,64 --causal --out fp32 | tee $O/ab_cfg2.json | cut -c1-400
$AB --shape 4,16,1024,64 --causal | tee -a $O/ab_cfg2.json | cut -c1-400
$AB --shape 2,16,1024,128 --causal | tee -a $O/ab_cfg2.json | cut -c1-400
$AB --shape 8,8,512,128 | tee -a $O/ab_cfg2.json | cut -c1-400
$AB --shape 1,24,4096,128 --out fp32 --inner 20 | tee $O/ab_flux.json | cut -c1-400
for sh in 1,16,4096,128 1,32,4096,128 1,24,4096,128 1,8,8192,128 1,12,8192,128 1,16,8192,128 1,4,16384,128; do
  python tools/ab_inproc.py --graph --rounds 10 --inner 20 --out fp32 --shape $sh new=intree | tee -a $O/rounds_probe.jsonl | cut -c1-300
done
timeout 600 python -m pytest tests/test_gpu_wide_heads.py -q -x 2>&1 | tail -8 | cut -c1-250
