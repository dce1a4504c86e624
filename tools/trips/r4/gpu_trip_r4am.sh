#!/bin/bash
O=gpurun_out/r4am; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/lab/window_probe.py > $O/window_probe.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4am/window_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'])
PY
tail -2 $O/err.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_w64.py tests/test_gpu_configs.py -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-300
