#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_w64_masks.py tests/test_gpu_forward.py -x -q > $O/tests_masks.txt 2>&1; tail -5 $O/tests_masks.txt | cut -c1-300
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_stamps.so UMFA_STRIP_TRUE_MASKS=0 timeout 600 python tools/lab/mask_stamps.py > $O/mask_stamps.txt 2>&1; cat $O/mask_stamps.txt
timeout 1200 python tools/lab/mask_w64_probe.py > $O/mask_w64_probe.jsonl 2>$O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4l/mask_w64_probe.jsonl'):
    d=json.loads(l)
    if 'w64_ms' in d: print(d['shape'], d['mask'][:14], d['w64_ms'], d['r128_ms'], d['r128_over_w64'])
    else: print(d['shape'], 'none', d['ms'])
PY
tail -2 $O/probe_err.txt
