#!/bin/bash
# round 5, trip l: the cost-model dispatcher -- routing test, fresh random probe (how far behind the best is the model's choice on shapes it was not fitted on)
O=gpurun_out/r5l; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_routing.py -q -x 2>&1 | tail -8 | cut -c1-600
for s in 201 202; do timeout 900 python tools/lab/routing_random_probe.py $s 150 > $O/routing_random_bf16_seed$s.jsonl 2>> $O/err.txt; tail -1 $O/routing_random_bf16_seed$s.jsonl; done
timeout 900 python tools/lab/routing_random_probe.py 203 120 fp16 > $O/routing_random_fp16_seed203.jsonl 2>> $O/err.txt; tail -1 $O/routing_random_fp16_seed203.jsonl
grep -h MISS $O/routing_random_*.jsonl | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['default_us'], d['default_kernel'].replace('fa_fwd16',''), 'w64', d['w64_us'], 'r128', d['r128_us'], d['default_over_best'])
"
