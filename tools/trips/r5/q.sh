#!/bin/bash
# round 5, trip q: XCD-weighted slices of the shared steps: potential, measured with this board's own clocks
O=gpurun_out/r5q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_w64_masks.py -q -x 2>&1 | tail -3 | cut -c1-300
python tools/lab/xcd_balance_probe.py 2>&1 | grep -v amdgpu | tee $O/xcd_balance_probe.txt
