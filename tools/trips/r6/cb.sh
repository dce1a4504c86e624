#!/bin/bash
# trip cb: soak of all 21 fuzz legs at the round's last build (ragged / unaligned masks on both kernels), 2500 fresh seeds
O=gpurun_out/r6cb; mkdir -p $O
(time timeout 3000 python3 tools/lab/value_fuzz.py 600000 2500) 2>&1 | tail -25 | tee $O/soak_2500_seeds_all_legs_last_build.txt
