#!/bin/bash
# round 4, trip J: mask kernel with the rolling list window and the 16-byte pack pass: parity, A/B against the 128-row kernel
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_w64_masks.py tests/test_gpu_forward.py -x -q > $O/tests_masks.txt 2>&1; tail -5 $O/tests_masks.txt | cut -c1-300
timeout 1200 python tools/lab/mask_w64_probe.py > $O/mask_w64_probe.jsonl 2>$O/probe_err.txt; cat $O/mask_w64_probe.jsonl | cut -c1-330; tail -3 $O/probe_err.txt
