#!/bin/bash
# trip ca: the realign tests after the tolerance between the kernel's two mask-read paths was set (each is checked against fp64)
O=gpurun_out/r6ca; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_f32_mask.py tests/test_gpu_forward.py -q 2>&1 | tail -5 | tee $O/tests.txt
