#!/bin/bash
# round 4, trip Y: fp8 P V kernel with P held at 2^5 x (normal range of e4m3): parity + time
O=gpurun_out/r4y; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fp8pv.py tests/test_gpu_quantized.py -m gpu -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4y/bench.json').read().strip().split('\n')[-1])
for k,v in d['int8'].items(): print(k, v['bf16_ms'], v['int8_ms_incl_quantiser'], v['speedup'], v['fp8pv_ms_incl_quantiser'], v['fp8pv_speedup'])
print(json.dumps(d['parity']['cfg4_fp8pv_B1_H16_S8192']))
PY
