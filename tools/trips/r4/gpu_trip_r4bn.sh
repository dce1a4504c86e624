#!/bin/bash
O=gpurun_out/r4bn; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/lab/routing_random_probe.py 9 120 > $O/routing_random_bf16_seed9.jsonl 2>> $O/err.txt
timeout 1500 python tools/lab/routing_random_probe.py 11 150 > $O/routing_random_bf16_seed11.jsonl 2>> $O/err.txt
timeout 1500 python tools/lab/routing_random_probe.py 12 100 fp16 > $O/routing_random_fp16_seed12.jsonl 2>> $O/err.txt
python3 - <<'PY'
import json
for f in ('bf16_seed9','bf16_seed11','fp16_seed12'):
    for l in open('gpurun_out/r4bn/routing_random_%s.jsonl' % f):
        d=json.loads(l)
        if d.get('MISS') and d['default_over_best'] > 1.08: print(f, d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'], d['default_over_best'])
        if 'launches' in d: print(f, d)
PY
