#!/bin/bash
O=gpurun_out/r4ap; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-300
timeout 900 python tools/lab/value_fuzz.py 15000 150 run_big_case > $O/fuzz_big.txt 2>&1; tail -1 $O/fuzz_big.txt
timeout 900 python tools/lab/value_fuzz.py 16000 300 run_shape_case > $O/fuzz_shape.txt 2>&1; tail -1 $O/fuzz_shape.txt
timeout 900 python tools/lab/value_fuzz.py 17000 200 run_case > $O/fuzz_case.txt 2>&1; tail -1 $O/fuzz_case.txt
