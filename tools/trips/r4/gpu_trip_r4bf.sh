#!/bin/bash
O=gpurun_out/r4bf; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/lab/routing_random_probe.py 5 150 > $O/routing_random_bf16_seed5.jsonl 2>> $O/err.txt
timeout 1500 python tools/lab/routing_random_probe.py 6 100 fp16 > $O/routing_random_fp16_seed6.jsonl 2>> $O/err.txt
python3 - <<'PY'
import json
for f in ('bf16_seed5','fp16_seed6'):
    for l in open('gpurun_out/r4bf/routing_random_%s.jsonl' % f):
        d=json.loads(l)
        if d.get('MISS') and d['default_over_best'] > 1.08: print(f, d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'], d['default_over_best'])
        if 'launches' in d: print(f, d)
PY
