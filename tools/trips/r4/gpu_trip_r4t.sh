#!/bin/bash
# round 4, trip T: mask kernel for fewer blocks than CUs (every block shared): probe against the 128-row kernel, mask tests
O=gpurun_out/r4t; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_w64_masks.py -m gpu -q > $O/tests_masks.txt 2>&1; tail -3 $O/tests_masks.txt | cut -c1-300
timeout 900 python tools/lab/mask_w64_probe.py few > $O/mask_w64_few_blocks.jsonl 2> $O/probe_err.txt; python3 - <<'PY'
import json
for l in open('gpurun_out/r4t/mask_w64_few_blocks.jsonl'):
    d=json.loads(l)
    print(d['shape'], d['mask'], d.get('visible'), d.get('w64_ms', d.get('ms')), d.get('r128_ms'), d.get('r128_over_w64'), d.get('max_rel_diff'), d.get('w64_kernel', d.get('kernel')))
PY
tail -3 $O/probe_err.txt | cut -c1-300
