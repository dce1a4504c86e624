#!/bin/bash
# round 4, trip AQ: HEAD: whole GPU suite + smoke + the driver's bench command; kernel trace of the bench (headline-only, eager)
O=gpurun_out/r4aq; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4aq/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['warmup'])
for k,v in d['int8'].items(): print(k, v['bf16_ms'], v['int8_ms_incl_quantiser'], v['speedup'], v['fp8pv_ms_incl_quantiser'], v['fp8pv_speedup'])
print({k:(v.get('ms'),v.get('frac'),v.get('rel')) for k,v in d['configs'].items()})
print({k:v.get('rel') for k,v in d['parity'].items() if isinstance(v,dict)})
PY
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof.json 2>$R/$O/prof_err.txt )
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fwd_kernel_stats.csv; cut -c1-200 $O/fwd_kernel_stats.csv | head -4
rm -rf $O/trace; find $O -name "*.db" -delete
