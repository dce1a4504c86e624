#!/bin/bash
# round 3, trip I: slice skew of two-way cut items, GQA group sum vectorised
O=gpurun_out/r3i; mkdir -p $O
timeout 600 python tools/lab/skew_probe.py > $O/skew.json 2>$O/skew_err.txt; cat $O/skew.json; tail -3 $O/skew_err.txt
timeout 300 python tools/bench_gqa_train.py > $O/gqa_train.json 2>$O/gqa_err.txt; cat $O/gqa_train.json
timeout 900 python -m pytest tests/test_gpu_sdpa.py tests/test_gpu_backward.py tests/test_gpu_w64.py -q -x > $O/tests_a.txt 2>&1; tail -4 $O/tests_a.txt
