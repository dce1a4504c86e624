#!/bin/bash
# round 4, trip AE: routing of bf16 calls with few q-blocks per head to the 128-row kernel; full suite; probes
O=gpurun_out/r4ae; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -6 $O/tests.txt | cut -c1-300
timeout 900 python tools/lab/small_nqb_probe.py > $O/small_nqb_probe_after.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ae/small_nqb_probe_after.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['default_kernel'], d['r128_us'], d['r128_over_default'])
PY
