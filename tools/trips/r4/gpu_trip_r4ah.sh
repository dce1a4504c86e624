#!/bin/bash
# round 4, trip AH: routing audit: the dispatcher's choice against the 128-row kernel over a grid of launch sizes (bf16, default options)
O=gpurun_out/r4ah; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/lab/small_nqb_probe.py sweep > $O/routing_sweep.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ah/routing_sweep.jsonl'):
    d=json.loads(l)
    flag = '' if d['r128_over_default'] >= 0.97 else '   <<<<'
    print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'], flag)
PY
