#!/bin/bash
# round 3, trip P: sliding windows on the one-wave-per-SIMD structure -- parity, then timing against the 128-row kernel
O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w64.py -x -q -k "window" > $O/window_tests.txt 2>&1; tail -25 $O/window_tests.txt
timeout 600 python tools/lab/window_probe.py > $O/window_probe.jsonl 2>$O/err.txt; cat $O/window_probe.jsonl; tail -3 $O/err.txt
timeout 1500 python -m pytest tests/test_gpu_w64.py tests/test_gpu_forward.py tests/test_gpu_configs.py -x -q > $O/fwd_tests.txt 2>&1; tail -5 $O/fwd_tests.txt
