#!/bin/bash
O=gpurun_out/r4bd; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/lab/split_gap_probe.py > $O/split_gap_probe.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4bd/split_gap_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['items128'], 'plan', d['plan_us'], 'x2', d['two_parts_us'], 'x3', d['three_parts_us'], 'w64', d['w64_us'])
PY
tail -2 $O/err.txt | cut -c1-200
