#!/bin/bash
O=gpurun_out/r4bk; mkdir -p $O
export TMPDIR=/tmp
true
timeout 1800 python tools/lab/split_plan_random.py 1 60 > $O/split_plan_random.jsonl 2> $O/err.txt
timeout 1800 python tools/lab/split_plan_random.py 2 60 > $O/split_plan_random_seed2.jsonl 2>> $O/err.txt
python3 - <<'PY'
import json
for f in ('split_plan_random','split_plan_random_seed2'):
    for l in open('gpurun_out/r4bk/%s.jsonl' % f):
        d=json.loads(l)
        if d.get('MISS'): print(d['shape'], d['items'], 'plan', d['plan_us'], 'k1', d['k1_us'], {k:v for k,v in d.items() if k.startswith('k') and k.endswith('_us') and k!='k1_us'}, 'best', d['best_k_upto8'], d['plan_over_best'])
        if 'launches' in d: print(f, d)
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
