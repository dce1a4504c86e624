#!/bin/bash
O=gpurun_out/r4bh; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python tools/lab/routing_random_masks.py 1 80 > $O/routing_random_masks.jsonl 2> $O/err.txt
python3 - <<'PY'
import json
for l in open('gpurun_out/r4bh/routing_random_masks.jsonl'):
    d=json.loads(l)
    if d.get('MISS'): print(d['shape'], d['mask'], d['visible'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], d['w64_kernel'].replace('fa_fwd16',''), 'r128', d['r128_us'], d['default_over_best'])
    if 'launches' in d: print(d)
PY
tail -2 $O/err.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_w64_masks.py tests/test_gpu_value_fuzz.py -m gpu -q 2>&1 | tail -2
