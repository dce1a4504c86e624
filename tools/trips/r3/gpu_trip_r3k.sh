#!/bin/bash
# round 3, trip K: head_dim 64 on the one-wave-per-SIMD structure -- parity first, then the A/B against the 128-row kernel
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w64.py -x -q -k "head_dim_64" > $O/d64_tests.txt 2>&1; tail -15 $O/d64_tests.txt
timeout 900 python tools/lab/d64_probe.py > $O/d64_probe.jsonl 2>$O/d64_probe_err.txt; cat $O/d64_probe.jsonl; tail -5 $O/d64_probe_err.txt
timeout 1200 python -m pytest tests/test_gpu_w64.py tests/test_gpu_configs.py -x -q > $O/w64_tests.txt 2>&1; tail -5 $O/w64_tests.txt
