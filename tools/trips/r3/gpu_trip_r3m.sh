#!/bin/bash
# round 3, trip M: head_dim 64 is vector-issue bound -- do row sums on the (half idle) matrix pipe pay THERE?
O=gpurun_out/r3m; mkdir -p $O
for sh in "2,16,4096,64" "1,16,8192,64"; do
  timeout 600 python tools/ab_inproc.py --shape $sh --rounds 14 --inner 20 ctl=tools/lab_bin/libMFAFFI_ctl.so msum=tools/lab_bin/libMFAFFI_msum.so >> $O/ab_msum_d64.jsonl 2>>$O/ab_err.txt
done
timeout 600 python tools/ab_inproc.py --shape 4,16,4096,64 --causal --rounds 10 --inner 10 ctl=tools/lab_bin/libMFAFFI_ctl.so msum=tools/lab_bin/libMFAFFI_msum.so >> $O/ab_msum_d64.jsonl 2>>$O/ab_err.txt
timeout 600 python tools/ab_inproc.py --shape 2,16,4096,64 --dtype fp16 --rounds 10 --inner 10 ctl=tools/lab_bin/libMFAFFI_ctl.so msum=tools/lab_bin/libMFAFFI_msum.so >> $O/ab_msum_d64.jsonl 2>>$O/ab_err.txt
cat $O/ab_msum_d64.jsonl; tail -3 $O/ab_err.txt
timeout 900 python -m pytest tests/test_gpu_w64.py -x -q -k "dispatch_gate" > $O/gate_tests.txt 2>&1; tail -3 $O/gate_tests.txt
