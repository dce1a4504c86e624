#!/bin/bash
# round 5, trip a: the range-safe default forward -- whole GPU suite, smoke, bench, kernel trace
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pv16_range.py -q -x > $O/tests_range.txt 2>&1; tail -15 $O/tests_range.txt | cut -c1-300
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -30 $O/tests.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5a/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'])
for k,v in d.get('configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('ms','rel','frac','kernel','frac_of_visible_work','speedup')})
print('int8', json.dumps(d.get('int8'))[:600])
print('parity', json.dumps(d.get('parity'))[:800])
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {} | cut -c1-200'
