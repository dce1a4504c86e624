#!/bin/bash
# round 4, trip R: mask kernel with cut blocks (stream-K over tile lists): tests, probe against the 128-row kernel
O=gpurun_out/r4r; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_w64_masks.py -m gpu -q > $O/tests_masks.txt 2>&1; tail -5 $O/tests_masks.txt | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_w64_masks.py > $O/tests_rest.txt 2>&1; tail -4 $O/tests_rest.txt | cut -c1-300
