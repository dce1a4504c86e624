#!/bin/bash
# round 4, trip AV: split-KV fold in two passes (independent loads): decode-like shapes, split probe, tests
O=gpurun_out/r4av; mkdir -p $O
export TMPDIR=/tmp
for a in "1 32 1 8192 128" "1 32 1 32768 128" "8 32 1 8192 128" "1 8 1 131072 128" "1 4 1 65536 128" "1 2 1 131072 128" "1 4 1 8192 128" "16 16 1 2048 64" "4 32 16 8192 128" "1 32 128 8192 128"; do timeout 60 python tools/bench_decode.py $a 2>/dev/null | tail -1; done | tee $O/decode.txt
timeout 900 python tools/lab/split_probe.py > $O/split_probe.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4av/split_probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['default_us'], d['no_split_us'], d.get('causal_half_split_us'), d.get('default_vs_no_split_rel'))
PY
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -2
