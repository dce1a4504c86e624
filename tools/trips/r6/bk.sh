#!/bin/bash
# trip bk: bf16 additive masks -- the classification pass at ~10 vector instructions per element (was 14), one workgroup per (256-row block, key tile), the fp16 copy
# written only where the bias kernel reads it -- tests, the mask fuzz leg, in-process A/B against the library before the change
O=gpurun_out/r6bk; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_bias.py tests/test_gpu_w64_f32_mask.py tests/test_gpu_w64_masks.py tests/test_gpu_value_fuzz.py -q 2>&1 | tail -8 | tee $O/tests.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 70000 2500 run_w64_mask_case) 2>&1 | tail -6 | tee $O/fuzz_w64_mask_leg_2500_seeds.txt
for m in bias_bf16 blockdiag_bf16 bias bias_f32; do
  timeout 300 python3 tools/ab_inproc.py --graph --out fp32 --mask $m before=tools/lab/tmp_old/libMFAFFI_before_bf16_pass.so after=intree 2>&1 | tail -4 | tee -a $O/ab_bf16_pass.txt
done
timeout 300 python3 tools/ab_inproc.py --graph --out fp32 --shape 2,16,4096,64 --mask bias_bf16 before=tools/lab/tmp_old/libMFAFFI_before_bf16_pass.so after=intree 2>&1 | tail -4 | tee -a $O/ab_bf16_pass.txt
