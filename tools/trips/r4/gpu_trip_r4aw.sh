#!/bin/bash
O=gpurun_out/r4aw; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python tools/lab/bwd_routing_probe.py > $O/bwd_routing_probe.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4aw/bwd_routing_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['dq1_us'], d['dq2_us'], d['persist_us'], d['default_over_best'])
PY
tail -2 $O/err.txt | cut -c1-200
