#!/bin/bash
# round 4, trip S: mask kernel with cut blocks, after the fewer-steps-than-workgroups fix: mask tests (3 times: a race shows up as flakiness), probe
O=gpurun_out/r4s; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_w64_masks.py -m gpu -q > $O/tests_masks_$i.txt 2>&1; tail -2 $O/tests_masks_$i.txt | cut -c1-300; done
timeout 900 python tools/lab/mask_w64_probe.py > $O/mask_w64_vs_128row.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4s/mask_w64_vs_128row.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['mask'], d.get('visible'), d.get('w64_ms', d.get('ms')), d.get('r128_ms'), d.get('r128_over_w64'), d.get('max_rel_diff'))
PY
