#!/bin/bash
# round 4, trip AI: slightly fewer items than CUs: stream-K against whole items
O=gpurun_out/r4ai; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/lab/grid_probe.py more > $O/grid_probe.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ai/grid_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['items'], d['streamk_us'], d['whole_us'], d['r128_us'])
PY
tail -2 $O/probe_err.txt | cut -c1-200
