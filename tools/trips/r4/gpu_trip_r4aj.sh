#!/bin/bash
# round 4, trip AJ: after the grid rule: both grid probes (evidence files), the routing sweep again, full suite
O=gpurun_out/r4aj; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/lab/grid_probe.py > $O/grid_probe.jsonl 2> $O/err1.txt
timeout 600 python tools/lab/grid_probe.py more > $O/grid_probe_more.jsonl 2> $O/err2.txt
timeout 1500 python tools/lab/small_nqb_probe.py sweep > $O/routing_sweep.jsonl 2> $O/err3.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4aj/routing_sweep.jsonl'):
    d=json.loads(l)
    if d['r128_over_default'] < 0.97 or 'full' in d['shape'] and 'D128' in d['shape']: print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), d['r128_us'], d['r128_over_default'], '' if d['r128_over_default'] >= 0.97 else '   <<<<')
PY
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
