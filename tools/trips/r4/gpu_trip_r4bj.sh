#!/bin/bash
O=gpurun_out/r4bj; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python tools/lab/decode_k_probe.py > $O/decode_k_probe.jsonl 2> $O/err.txt
python3 - <<'PY'
import json
for l in open('gpurun_out/r4bj/decode_k_probe.jsonl'):
    d=json.loads(l)
    ks={k:v for k,v in d.items() if k.startswith('k')}
    best=min(ks,key=ks.get)
    print(d['shape'], 'items', d['items'], 'ntiles', d['ntiles'], 'plan', d['plan'], 'best', best, ks[best], ks)
PY
tail -2 $O/err.txt | cut -c1-300
