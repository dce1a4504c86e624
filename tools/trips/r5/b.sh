#!/bin/bash
# round 5, trip b: cast pass with relaxed atomics -- range tests, streams test, headline bench + kernel trace
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pv16_range.py tests/test_gpu_streams.py tests/test_gpu_w64.py -q -x > $O/tests_range.txt 2>&1; tail -15 $O/tests_range.txt | cut -c1-300
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof.json 2>$R/$O/prof_err.txt )
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -6 {} | cut -c1-220'
timeout 900 python bench.py --steps 20 --warmup 5 --headline-only > $O/bench_headline.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5b/bench_headline.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'], d.get('settle',{}).get('cold_start_ms_per_step'))
PY
