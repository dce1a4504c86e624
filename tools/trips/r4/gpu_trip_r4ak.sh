#!/bin/bash
# round 4, trip AK: gates after the audit: full suite, the sweep's former misses, bench
O=gpurun_out/r4ak; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
timeout 1500 python tools/lab/small_nqb_probe.py sweep > $O/routing_sweep.jsonl 2> $O/err3.txt; python3 - <<'PY'
import json
n=0
for l in open('gpurun_out/r4ak/routing_sweep.jsonl'):
    d=json.loads(l); n+=1
    if d['r128_over_default'] < 0.97: print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'], '   <<<<')
print(n, 'shapes')
PY
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4ak/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k:(v.get('ms'),v.get('frac')) for k,v in d['configs'].items()})
PY
