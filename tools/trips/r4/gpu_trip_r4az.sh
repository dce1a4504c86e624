#!/bin/bash
O=gpurun_out/r4az; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/lab/small_nqb_probe.py sweep fp16 > $O/routing_sweep_fp16.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
n=0
for l in open('gpurun_out/r4az/routing_sweep_fp16.jsonl'):
    d=json.loads(l); n+=1
    if d['r128_over_default'] < 0.97: print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'], '   <<<<')
print(n, 'shapes')
PY
timeout 600 python tools/lab/small_nqb_probe.py fp16 > $O/small_nqb_fp16.jsonl 2>> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4az/small_nqb_fp16.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'])
PY
