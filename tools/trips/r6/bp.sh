#!/bin/bash
# trip bp: the backward fuzz leg with its second reference (D from the rounded O): the failing seed, then 3000 seeds of the leg
O=gpurun_out/r6bp; mkdir -p $O
timeout 300 python3 tools/lab/value_fuzz.py 401643 1 run_bwd_case 2>&1 | tail -3 | tee $O/bwd_leg_seed_401643.txt
(time timeout 1500 python3 tools/lab/value_fuzz.py 500000 3000 run_bwd_case) 2>&1 | tail -8 | tee $O/fuzz_bwd_leg_3000_seeds.txt
