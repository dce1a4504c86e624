#!/bin/bash
O=gpurun_out/r4bo; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
timeout 1500 python tools/lab/routing_random_probe.py 12 100 fp16 > $O/routing_random_fp16_seed12.jsonl 2>> $O/err.txt
timeout 1500 python tools/lab/routing_random_probe.py 13 120 > $O/routing_random_bf16_seed13.jsonl 2>> $O/err.txt
python3 - <<'PY'
import json
for f in ('fp16_seed12','bf16_seed13'):
    for l in open('gpurun_out/r4bo/routing_random_%s.jsonl' % f):
        d=json.loads(l)
        if d.get('MISS') and d['default_over_best'] > 1.08: print(f, d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'], d['default_over_best'])
        if 'launches' in d: print(f, d)
PY
for dt in bf16 fp16; do timeout 1500 python tools/lab/small_nqb_probe.py sweep $dt > $O/routing_sweep_$dt.jsonl 2>> $O/err.txt; python3 - $dt <<'PY'
import json,sys
n=0
for l in open('gpurun_out/r4bo/routing_sweep_%s.jsonl' % sys.argv[1]):
    d=json.loads(l); n+=1
    if d['r128_over_default'] < 0.97: print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'], '   <<<<')
print(sys.argv[1], n, 'sweep shapes')
PY
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
