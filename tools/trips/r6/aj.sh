#!/bin/bash
# trip aj: the decode form of the 128-row kernel (KS = 4): tests, then decode shapes with the form on / off
O=gpurun_out/r6aj; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_decode.py -m gpu -q -x 2>&1 | tail -15 | tee $O/tests.txt
for sh in "1 32 1 8192 128" "8 32 1 8192 128" "1 32 1 32768 128" "32 32 1 2048 128" "8 32 1 8192 64" "4 32 8 8192 128" "16 8 1 4096 128" "1 8 1 131072 128" "64 8 1 1024 128" "1 8 4 8192 128" "2 8 32 4096 128"; do
  for m in 0 2; do echo -n "decode_ks=$m  "; UMFA_DECODE_KS=$m timeout 120 python3 tools/bench_decode.py $sh 2>&1 | tail -1; done
done | tee $O/decode_form.txt
