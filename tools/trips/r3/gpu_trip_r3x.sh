#!/bin/bash
# round 3, trip X: lazy softmax reference with fp16 P (fp16 kernels + the int8 kernel): parity, then lazy vs deferred timing
O=gpurun_out/r3x; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_w64.py tests/test_gpu_quantized.py tests/test_gpu_fp8pv.py tests/test_gpu_configs.py tests/test_gpu_forward.py -x -q > $O/tests.txt 2>&1; tail -12 $O/tests.txt
timeout 900 python tools/lab/lazy16_probe.py > $O/lazy16.jsonl 2>$O/err.txt; cat $O/lazy16.jsonl; tail -3 $O/err.txt
