#!/bin/bash
O=gpurun_out/r4bc; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/lab/routing_random_probe.py 1 120 > $O/routing_random_bf16.jsonl 2> $O/err.txt
timeout 1500 python tools/lab/routing_random_probe.py 2 60 fp16 > $O/routing_random_fp16.jsonl 2>> $O/err.txt
python3 - <<'PY'
import json
for f in ('bf16','fp16'):
    for l in open('gpurun_out/r4bc/routing_random_%s.jsonl' % f):
        d=json.loads(l)
        if d.get('MISS'): print(f, d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'], d['default_over_best'])
        if 'launches' in d: print(f, d)
PY
tail -2 $O/err.txt | cut -c1-200
