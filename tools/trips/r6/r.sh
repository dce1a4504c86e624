#!/bin/bash
# trip r: balanced causal pairs (CBAL) on the 128-row kernel: parity + timing against the unpaired schedule, cbal_delta 0..3
O=gpurun_out/r6r; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for d in 0 1 2 3; do cp $L tools/lab_bin/libMFAFFI_cb$d.so; done
for sh in 4,16,1024,64 8,16,512,64 2,16,2048,64 4,8,1024,128 2,16,1024,128 1,16,2048,128 4,16,256,64; do
timeout 300 python3 tools/ab_inproc.py --shape $sh --causal --out fp32 --graph --inner 100 --rounds 10 --parity "off=$L:cbal=2" "d0=tools/lab_bin/libMFAFFI_cb0.so:cbal_delta=0" "d1=tools/lab_bin/libMFAFFI_cb1.so:cbal_delta=1" "d2=tools/lab_bin/libMFAFFI_cb2.so:cbal_delta=2" "d3=tools/lab_bin/libMFAFFI_cb3.so:cbal_delta=3" 2>&1 | grep -E "shape|Error|error|assert" | tee -a $O/ab_cbal.jsonl
done
# forced beyond one resident round
for sh in 4,16,2048,64 8,16,1024,64 2,16,4096,64; do
timeout 300 python3 tools/ab_inproc.py --shape $sh --causal --out fp32 --graph --inner 50 --rounds 8 --parity "off=$L:cbal=2,no_w64=1" "on1=tools/lab_bin/libMFAFFI_cb1.so:cbal=1,cbal_delta=1,no_w64=1" "on2=tools/lab_bin/libMFAFFI_cb2.so:cbal=1,cbal_delta=2,no_w64=1" 2>&1 | grep -E "shape|Error|error|assert" | tee -a $O/ab_cbal_forced.jsonl
done
timeout 900 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_pv16_range.py -m gpu -x -q 2>&1 | tail -15 | tee $O/tests.txt
