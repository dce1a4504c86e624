#!/bin/bash
# round 4, trip E: four-slot LDS-DMA ring at head_dim <= 64 in the 128-row kernel (config 2): parity, stamps, bench
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -8 $O/tests.txt | cut -c1-250
for pv in 0 1; do timeout 120 tools/lab_bin/cfg2_lab_ns4_pv$pv 16 1024 50 4 > $O/cfg2_stamps_ns4_pv$pv.txt 2>&1; tail -4 $O/cfg2_stamps_ns4_pv$pv.txt; done
timeout 600 python tools/bench_cfg2.py > $O/bench_cfg2.json 2>$O/cfg2_err.txt; cat $O/bench_cfg2.json; tail -2 $O/cfg2_err.txt
timeout 600 python tools/bench_mask.py > $O/bench_mask.txt 2>&1; tail -20 $O/bench_mask.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4e/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['int8'].items(): print(k, v['bf16_ms'], v['int8_ms_incl_quantiser'], v['speedup'], v['fp8pv_ms_incl_quantiser'], v['fp8pv_speedup'])
print({k:(v.get('ms'),v.get('frac')) for k,v in d['configs'].items()})
PY
