#!/bin/bash
# round 4, trip AC: fa_fwd16 split-KV fold without fences: tests, probe
O=gpurun_out/r4ac; mkdir -p $O
export TMPDIR=/tmp
echo skip tests
timeout 900 python tools/lab/split_probe.py > $O/split_probe.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ac/split_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['no_split_us'], d.get('causal_half_split_us'), d.get('default_vs_no_split_rel'), d.get('causal_half_split_vs_no_split_rel'))
PY
tail -2 $O/probe_err.txt | cut -c1-300
