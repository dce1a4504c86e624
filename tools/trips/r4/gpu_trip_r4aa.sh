#!/bin/bash
# round 4, trip AA: soak at HEAD -- value fuzz, fresh seeds (every leg), then the mask leg and the big-shape leg longer
O=gpurun_out/r4aa; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/lab/value_fuzz.py 7000 120 > $O/fuzz_all.txt 2>&1; tail -3 $O/fuzz_all.txt | cut -c1-300
timeout 600 python tools/lab/value_fuzz.py 8000 400 run_mask_case > $O/fuzz_mask.txt 2>&1; tail -2 $O/fuzz_mask.txt | cut -c1-300
timeout 600 python tools/lab/value_fuzz.py 9000 60 run_big_case > $O/fuzz_big.txt 2>&1; tail -2 $O/fuzz_big.txt | cut -c1-300
