#!/bin/bash
O=gpurun_out/r4ba; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-250
timeout 1500 python tools/lab/small_nqb_probe.py sweep fp16 > $O/routing_sweep_fp16.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
n=0
for l in open('gpurun_out/r4ba/routing_sweep_fp16.jsonl'):
    d=json.loads(l); n+=1
    if d['r128_over_default'] < 0.97: print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'], '   <<<<')
print(n, 'shapes')
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
