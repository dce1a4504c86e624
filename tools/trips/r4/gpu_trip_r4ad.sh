#!/bin/bash
# round 4, trip AD: few q-blocks per head: one-wave-per-SIMD kernel + V cast pass against the 128-row kernel with in-kernel conversion
O=gpurun_out/r4ad; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/lab/small_nqb_probe.py > $O/small_nqb_probe.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ad/small_nqb_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['default_kernel'], d['r128_us'], d['r128_over_default'])
PY
tail -2 $O/probe_err.txt | cut -c1-300
