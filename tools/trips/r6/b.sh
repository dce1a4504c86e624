#!/bin/bash
# trip b: additive fp16 masks on the one-wave-per-SIMD structure (MASKA), first build: parity tests + timing against the 128-row kernel
O=gpurun_out/r6b; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_w64_bias.py -x -q 2>&1 | tail -15 | tee $O/tests.txt
timeout 600 python3 tools/lab/bias_probe.py 2>&1 | grep -v amdgpu | tee $O/bias_probe.jsonl
