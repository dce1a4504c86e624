#!/bin/bash
# trip bf: soak of all 21 fuzz legs at the fp32-mask build, 5000 fresh seeds
O=gpurun_out/r6bf; mkdir -p $O
(time timeout 3000 python3 tools/lab/value_fuzz.py 200000 5000) 2>&1 | tail -25 | tee $O/soak_5000_seeds_all_legs_fp32_mask_build.txt
