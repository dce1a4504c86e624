#!/bin/bash
# round 5, trip p: D64 mask tests; per-XCD end times on (probably) another box -- is the XCD speed order a property of the part or of the board?
O=gpurun_out/r5p; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_w64_masks.py -q -x 2>&1 | tail -5 | cut -c1-400
rocm-smi --showserial 2>/dev/null | grep -i serial | head -2
python tools/lab/w64_wg_times.py 1 24 4096 128 tools/lab_bin/libMFAFFI_stamps.so 2>/dev/null | head -10 | tee $O/wg_times.txt
python tools/lab/w64_wg_times.py 1 32 4096 128 tools/lab_bin/libMFAFFI_stamps.so 2>/dev/null | head -10 | tee -a $O/wg_times.txt
