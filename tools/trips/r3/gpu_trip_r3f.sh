#!/bin/bash
# round 3, trip F: causal half-split (config 2), capture-private scratch, in-stream quantised backward, full suite, bench
O=gpurun_out/r3f; mkdir -p $O
timeout 600 python tools/bench_cfg2.py > $O/cfg2.json 2>$O/cfg2_err.txt; cat $O/cfg2.json; tail -3 $O/cfg2_err.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -12 $O/gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_regime.json 2>$O/bench_err.txt
tail -c 300 $O/bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3f/bench_driver_regime.json'))
print(d['value'], d['ms_per_step'], d['settle']['cold_start_ms_per_step'], d['roofline']['frac'])
for k,v in d['configs'].items(): print(k, {a:b for a,b in v.items() if a in('ms','tflops','frac','kernel','rel','rms','error')})
PY
