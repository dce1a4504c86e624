#!/bin/bash
O=gpurun_out/r4ax; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python tools/lab/i8_routing_probe.py > $O/i8_routing_probe.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ax/i8_routing_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['items'], d['default_us'], d['default_kernel'], 'r128', d['r128_us'], 'cut256', d['cut256_us'])
PY
tail -2 $O/err.txt | cut -c1-200
