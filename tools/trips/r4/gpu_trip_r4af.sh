#!/bin/bash
# round 4, trip AF: the gate at Sq 1024 ... 2048 with the V cast pass in the picture
O=gpurun_out/r4af; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/lab/small_nqb_probe.py gate > $O/gate_probe.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4af/gate_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['default_kernel'], d['r128_us'], d['r128_over_default'])
PY
timeout 600 python -m pytest tests/test_gpu_w64.py -m gpu -q -k "dispatch_gate" > $O/tests.txt 2>&1; tail -2 $O/tests.txt | cut -c1-200
