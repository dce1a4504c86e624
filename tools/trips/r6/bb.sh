#!/bin/bash
# trip bb: fp32 additive masks through the guarded pair -- the new tests, the neighbouring mask suites, the mask fuzz leg, timing
O=gpurun_out/r6bb; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_f32_mask.py -q -x 2>&1 | tail -25 | tee $O/tests_f32_mask.txt
timeout 1200 python3 -m pytest tests/test_gpu_w64_bias.py tests/test_gpu_w64_masks.py tests/test_gpu_value_fuzz.py tests/test_gpu_forward.py -q 2>&1 | tail -8 | tee $O/tests_neighbours.txt
timeout 600 python3 tools/bench_mask_f32.py $O/mask_f32_timing.jsonl 2>&1 | tail -40
