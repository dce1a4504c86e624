#!/bin/bash
# round 4, trip AT: V cast pass with broadcast dimensions (zero-copy GQA): tests + GQA bench
O=gpurun_out/r4at; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sdpa.py tests/test_gpu_value_fuzz.py tests/test_gpu_library.py tests/test_gpu_smoke_entry.py -m gpu -q > $O/tests.txt 2>&1; tail -2 $O/tests.txt | cut -c1-200
