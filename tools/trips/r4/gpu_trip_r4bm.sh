#!/bin/bash
O=gpurun_out/r4bm; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4bm/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k:(v.get('ms'),v.get('frac'),v.get('kernel')) for k,v in d['configs'].items() if 'cfg5' in k or 'cfg2' in k})
PY
