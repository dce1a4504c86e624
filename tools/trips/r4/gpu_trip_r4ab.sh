#!/bin/bash
# round 4, trip AB: lazy tile bodies for key-padding masks: mask tests (twice), probe
O=gpurun_out/r4ab; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_w64_masks.py -m gpu -q > $O/tests_masks_$i.txt 2>&1; tail -3 $O/tests_masks_$i.txt | cut -c1-300; done
grep -n "^E  " $O/tests_masks_1.txt | head -8 | cut -c1-300
timeout 900 python tools/lab/mask_w64_probe.py > $O/mask_w64_vs_128row.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4ab/mask_w64_vs_128row.jsonl'):
    d=json.loads(l)
    if 'padding' in d['mask'] or d['mask']=='none': print(d['shape'], d['mask'], d.get('w64_ms', d.get('ms')), d.get('r128_ms'), d.get('r128_over_w64'), d.get('max_rel_diff'))
PY
