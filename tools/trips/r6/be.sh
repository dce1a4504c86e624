#!/bin/bash
# trip be: fp32 additive masks, fourth build (tile 0 of a block that sees nothing is written too) -- tests, neighbours, a soak of the mask fuzz leg
O=gpurun_out/r6be; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_f32_mask.py -q 2>&1 | tail -25 | tee $O/tests_f32_mask.txt
timeout 1200 python3 -m pytest tests/test_gpu_w64_bias.py tests/test_gpu_w64_masks.py tests/test_gpu_value_fuzz.py tests/test_gpu_forward.py tests/test_gpu_routing.py -q 2>&1 | tail -8 | tee $O/tests_neighbours.txt
(time timeout 2400 python3 tools/lab/value_fuzz.py 40000 3000 run_w64_mask_case) 2>&1 | tail -12 | tee $O/fuzz_w64_mask_leg_3000_seeds_with_fp32_masks.txt
