#!/bin/bash
# round 4, trip X: whole GPU suite + smoke + the driver's bench command at HEAD
O=gpurun_out/r4x; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4x/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['warmup'])
for k,v in d['int8'].items(): print(k, v['bf16_ms'], v['int8_ms_incl_quantiser'], v['speedup'], v['fp8pv_ms_incl_quantiser'], v['fp8pv_speedup'])
print({k:(v.get('ms'),v.get('frac'),v.get('rel')) for k,v in d['configs'].items()})
print({k:v.get('rel') for k,v in d['parity'].items() if isinstance(v,dict)})
PY
