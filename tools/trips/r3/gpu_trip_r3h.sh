#!/bin/bash
# round 3, trip H: GQA training in place (tests + time), full suite
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sdpa.py tests/test_gpu_backward.py -q -x > $O/tests_a.txt 2>&1; tail -8 $O/tests_a.txt
timeout 300 python tools/bench_gqa_train.py > $O/gqa_train.json 2>$O/gqa_err.txt; cat $O/gqa_train.json; tail -3 $O/gqa_err.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
