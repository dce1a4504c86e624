#!/bin/bash
# trip an: the suites that touch the slab exchanges, with the exchange never waiting (every early workgroup reads its slab itself) and with the two-launch forms
O=gpurun_out/r5an; mkdir -p $O
T="tests/test_gpu_pv16_range.py tests/test_gpu_w64.py tests/test_gpu_quantized.py tests/test_gpu_w64_masks.py tests/test_gpu_configs.py tests/test_gpu_value_fuzz.py tests/test_gpu_streams.py"
UMFA_CAST_WAIT_US=0 timeout 1500 python3 -m pytest $T -m gpu -q > $O/wait0.txt 2>&1; echo "cast_wait_us=0:"; tail -2 $O/wait0.txt | cut -c1-200
UMFA_CAST_TWO_PASS=1 timeout 1500 python3 -m pytest $T -m gpu -q > $O/twopass.txt 2>&1; echo "cast_two_pass=1:"; tail -2 $O/twopass.txt | cut -c1-200
