#!/bin/bash
# round 5, trip c: cast pass with flag-word exchange -- range tests, headline kernel trace for cast_u 4 / 16 / 32 / two-pass
O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pv16_range.py tests/test_gpu_w64.py -q -x > $O/tests_range.txt 2>&1; tail -5 $O/tests_range.txt | cut -c1-300
R=$GRAFT_REPO_ROOT
for u in 0 4 16 32 two; do
  if [ $u = two ]; then export UMFA_CAST_TWO_PASS=1; unset UMFA_CAST_U; else export UMFA_CAST_U=$u; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_u$u -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof_u$u.json 2>$R/$O/prof_err.txt )
  echo "== cast_u $u"; find $O/trace_u$u -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {} | cut -c1-150'
done
unset UMFA_CAST_TWO_PASS UMFA_CAST_U
timeout 900 python bench.py --steps 20 --warmup 5 --headline-only > $O/bench_headline.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c/bench_headline.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'], d.get('settle',{}).get('cold_start_ms_per_step'))
PY
