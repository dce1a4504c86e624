#!/bin/bash
# trip u: tests of the balanced causal pairs + the forward suites they touch
O=gpurun_out/r6u; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_cbal.py -m gpu -x -q 2>&1 | tail -25 | tee $O/cbal_tests.txt
timeout 1500 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_pv16_range.py tests/test_gpu_routing.py -m gpu -q 2>&1 | tail -25 | tee $O/touched_tests.txt
