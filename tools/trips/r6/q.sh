#!/bin/bash
# trip q: additive masks at head_dim 64 (Cfg(d64, madd)) + capture without a warm-up + fuzz of the mask kernels with head_dim 64 in the mix
O=gpurun_out/r6q; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_bias.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
timeout 900 python3 tools/lab/value_fuzz.py 5000 1200 run_w64_mask_case 2>&1 | grep -v amdgpu | tail -5 | tee $O/fuzz.txt
timeout 400 python3 tools/lab/bias_probe.py 2>&1 | grep -v amdgpu | grep ", 64\]" | tee $O/bias_probe_d64.jsonl
