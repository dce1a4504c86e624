#!/bin/bash
# trip bm: soak of all 21 fuzz legs at the round's final build (clean rebuild), 3000 fresh seeds
O=gpurun_out/r6bm; mkdir -p $O
(time timeout 3000 python3 tools/lab/value_fuzz.py 400000 3000) 2>&1 | tail -25 | tee $O/soak_3000_seeds_all_legs_final_build.txt
