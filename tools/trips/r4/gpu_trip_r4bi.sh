#!/bin/bash
O=gpurun_out/r4bi; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python tools/lab/split_plan_random.py 1 60 > $O/split_plan_random.jsonl 2> $O/err.txt
python3 - <<'PY'
import json
for l in open('gpurun_out/r4bi/split_plan_random.jsonl'):
    d=json.loads(l)
    if d.get('MISS'): print(d['shape'], d['items'], 'plan', d['plan_us'], 'k1', d['k1_us'], {k:v for k,v in d.items() if k.startswith('k') and k.endswith('_us') and k!='k1_us'}, 'best', d['best_k_upto8'], d['plan_over_best'])
    if 'launches' in d: print(d)
PY
tail -2 $O/err.txt | cut -c1-300
