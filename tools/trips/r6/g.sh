#!/bin/bash
# trip g: placement variants of the additive-mask bodies (where the eight mask DMA pieces sit, read-ahead depth, filler budget), in-process A/B at FLUX + fp16 bias
O=gpurun_out/r6g; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --mask bias --rounds 10 base=$L even=tools/lab_bin/libMFAFFI_m_even.so late=tools/lab_bin/libMFAFFI_m_late.so spread=tools/lab_bin/libMFAFFI_m_spread.so d1=tools/lab_bin/libMFAFFI_d1.so b32=tools/lab_bin/libMFAFFI_b32.so b36=tools/lab_bin/libMFAFFI_b36.so 2>&1 | grep shape | tee $O/ab_flux_bias.json
python3 tools/ab_inproc.py --shape 1,16,8192,128 --out fp32 --graph --mask bias --rounds 8 base=$L even=tools/lab_bin/libMFAFFI_m_even.so late=tools/lab_bin/libMFAFFI_m_late.so spread=tools/lab_bin/libMFAFFI_m_spread.so d1=tools/lab_bin/libMFAFFI_d1.so b32=tools/lab_bin/libMFAFFI_b32.so b36=tools/lab_bin/libMFAFFI_b36.so 2>&1 | grep shape | tee $O/ab_s8192_bias.json
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --mask bias_per_head --rounds 8 base=$L late=tools/lab_bin/libMFAFFI_m_late.so spread=tools/lab_bin/libMFAFFI_m_spread.so 2>&1 | grep shape | tee $O/ab_flux_bias_per_head.json
