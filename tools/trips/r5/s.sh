#!/bin/bash
# round 5, trip s: the quantised forward with the reference ABI's dense fp32 mask -- 16-byte mask loads in fa_fwd_i8
O=gpurun_out/r5s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_quantized.py -q -x 2>&1 | tail -3 | cut -c1-300
python tools/lab/i8_mask_probe.py 16 8192 2>&1 | grep -v amdgpu | tee $O/i8_mask_probe.txt
python tools/lab/i8_mask_probe.py 24 4096 2>&1 | grep -v amdgpu | tee -a $O/i8_mask_probe.txt
