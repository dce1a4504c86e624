#!/bin/bash
# trip f: bf16 additive masks (fp16 copy written by the classification pass) + the mask suites
O=gpurun_out/r6f; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_bias.py tests/test_gpu_w64_masks.py tests/test_gpu_forward.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
