#!/bin/bash
O=gpurun_out/r4ao; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/lab/bn32_probe.py > $O/bn32_probe.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ao/bn32_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['bn32_us'], d['bn64_us'], d['bn64_over_bn32'], d['rel_diff'])
PY
tail -2 $O/err.txt | cut -c1-300
