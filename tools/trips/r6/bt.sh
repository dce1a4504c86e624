#!/bin/bash
# trip bt: the mask fuzz legs at the shape-aware size rule
O=gpurun_out/r6bt; mkdir -p $O
(time timeout 1200 python3 tools/lab/value_fuzz.py 80000 2000 run_w64_mask_case) 2>&1 | tail -9 | tee $O/fuzz_w64_mask_leg_2000_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 80000 2000 run_mask_case) 2>&1 | tail -9 | tee $O/fuzz_mask_leg_2000_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 80000 1000 run_qmask_case) 2>&1 | tail -9 | tee $O/fuzz_qmask_leg_1000_seeds.txt
