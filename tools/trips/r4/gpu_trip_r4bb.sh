#!/bin/bash
O=gpurun_out/r4bb; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
for dt in bf16 fp16; do timeout 1500 python tools/lab/small_nqb_probe.py sweep $dt > $O/routing_sweep_$dt.jsonl 2> $O/err.txt; python3 - $dt <<'PY'
import json,sys
n=0
for l in open('gpurun_out/r4bb/routing_sweep_%s.jsonl' % sys.argv[1]):
    d=json.loads(l); n+=1
    if d['r128_over_default'] < 0.97: print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'], '   <<<<')
print(sys.argv[1], n, 'shapes')
PY
done
