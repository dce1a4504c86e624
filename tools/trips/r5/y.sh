#!/bin/bash
# trip y: the fp16 V image of the quantised forward as q * s * 2^-e -- range probe, the quantised / fuzz suites, int8 timing against the last commit's build
O=gpurun_out/r5y; mkdir -p $O
python3 tools/lab/qfwd_range_probe.py 2>&1 | grep -v amdgpu.ids > $O/probe.txt; grep -c "0.00e+00\|e-0[4-9]" $O/probe.txt; grep -v "0.00e+00" $O/probe.txt | head -20
python3 -m pytest tests/test_gpu_quantized.py tests/test_gpu_value_fuzz.py tests/test_gpu_streams.py tests/test_gpu_pv16_range.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for s in 1,24,4096,128 1,16,8192,128; do
python3 tools/ab_inproc.py --shape $s --out fp32 --graph --quant 2 r4=tools/lab_bin/libMFAFFI_r4.so new=universal-metal-flash-attention_amd/lib/libMFAFFI.so >> $O/ab_int8.txt 2>&1
done
grep shape $O/ab_int8.txt
