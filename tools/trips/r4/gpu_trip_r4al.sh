#!/bin/bash
O=gpurun_out/r4al; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/lab/few_items_probe.py > $O/few_items_probe_after_cap.jsonl 2> $O/err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4al/few_items_probe_after_cap.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['steps_per_cu'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'])
PY
