#!/bin/bash
# round 4, trip AY: HEAD after the last reverts: whole suite, smoke, bench
O=gpurun_out/r4ay; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default_args.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4ay/bench_default_args.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['steps'], d['warmup'], d['cpu_baseline']['value'])
PY
for a in "1 32 1 8192 128" "1 4 1 8192 128" "8 32 1 8192 128"; do timeout 60 python tools/bench_decode.py $a 2>/dev/null | tail -1; done
