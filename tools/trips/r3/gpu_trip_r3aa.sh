#!/bin/bash
# round 3, trip AA: bf16 operands with fp16 P V (option pv_fp16): parity, time, error against the other regimes
O=gpurun_out/r3aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w64.py -x -q -k "fp16_pv or pv16" > $O/tests.txt 2>&1; tail -8 $O/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/err.txt; tail -3 $O/err.txt
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3aa/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in d['configs'].items():
    if k.startswith('cfg3_flux') and 'rel' in v: print(k, v['ms'], v['frac'], v['rel'], v['rms'], v['kernel'])
PY
