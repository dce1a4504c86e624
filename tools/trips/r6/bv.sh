#!/bin/bash
# trip bw: (second build) additive masks of ragged shapes on the bias kernels (padded fp16 copy) -- the new tests, the mask suites, the mask fuzz leg with ragged shapes
O=gpurun_out/r6bw; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_bias.py -q -k "ragged" 2>&1 | tail -25 | tee $O/tests_ragged.txt
timeout 1500 python3 -m pytest tests/test_gpu_w64_bias.py tests/test_gpu_w64_f32_mask.py tests/test_gpu_w64_masks.py tests/test_gpu_forward.py -q 2>&1 | tail -6 | tee $O/tests_mask_suites.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 120000 2500 run_w64_mask_case) 2>&1 | tail -12 | tee $O/fuzz_w64_mask_leg_2500_seeds.txt
