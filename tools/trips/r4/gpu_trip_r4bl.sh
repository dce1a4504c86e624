#!/bin/bash
# round 4, trip BL: cost-model split plan: whole suite, decode / few-item probes, routing sweeps, smoke, bench
O=gpurun_out/r4bl; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
timeout 600 python tools/lab/decode_k_probe.py > $O/decode_k_probe.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4bl/decode_k_probe.jsonl'):
    d=json.loads(l)
    ks={k:v for k,v in d.items() if k.startswith('k')}
    best=min(ks,key=ks.get)
    print(d['shape'], 'plan', d['plan'], 'best', best, ks[best], round(d['plan']/ks[best],3))
PY
timeout 600 python tools/lab/few_items_probe.py > $O/few_items.jsonl 2>> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4bl/few_items.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'])
PY
timeout 1500 python tools/lab/routing_random_probe.py 9 120 > $O/routing_random_bf16_seed9.jsonl 2>> $O/err.txt
python3 - <<'PY'
import json
for l in open('gpurun_out/r4bl/routing_random_bf16_seed9.jsonl'):
    d=json.loads(l)
    if d.get('MISS') and d['default_over_best'] > 1.08: print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'], d['default_over_best'])
    if 'launches' in d: print(d)
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4bl/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
