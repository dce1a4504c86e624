#!/bin/bash
# round 5, trip d: whole GPU suite + smoke + full bench line at the range-safe build
O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -12 $O/tests.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5d/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'])
for k,v in d.get('configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('ms','rel','frac','kernel','frac_of_visible_work','speedup')})
print('int8', json.dumps(d.get('int8'))[:900])
PY
