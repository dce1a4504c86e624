#!/bin/bash
# round 4, trip AN: soak after the routing changes: value fuzz (fresh seeds, every leg), big-shape leg, full suite twice
O=gpurun_out/r4an; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python tools/lab/value_fuzz.py 11000 300 > $O/fuzz_all.txt 2>&1; tail -2 $O/fuzz_all.txt | cut -c1-300
timeout 1200 python tools/lab/value_fuzz.py 12000 200 run_big_case > $O/fuzz_big.txt 2>&1; tail -2 $O/fuzz_big.txt | cut -c1-300
timeout 1200 python tools/lab/value_fuzz.py 13000 400 run_shape_case > $O/fuzz_shape.txt 2>&1; tail -2 $O/fuzz_shape.txt | cut -c1-300
for i in 1 2; do timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_$i.txt 2>&1; tail -2 $O/tests_$i.txt | cut -c1-200; done
