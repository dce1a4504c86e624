#!/bin/bash
# trip aa: quantised backward with every operand as a power-of-two multiple -- probes, suites, timing against the build before
O=gpurun_out/r5aa; mkdir -p $O
python3 tools/lab/qbwd_range_probe.py 2>&1 | grep -v amdgpu.ids > $O/probe.txt; cat $O/probe.txt
python3 -m pytest tests/test_gpu_quantized.py tests/test_gpu_backward.py tests/test_gpu_value_fuzz.py tests/test_gpu_configs.py tests/test_gpu_sdpa.py tests/test_gpu_library.py tests/test_gpu_legacy_entries.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5aa/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('cfg3_flux_bf16_bwd','cfg4_int8_bwd','cfg3_flux_bf16_mask_blockdiag','cfg4_int8_mask_blockdiag'):
    print(k, d['configs'].get(k))
print(json.dumps(d['int8'])[:900])
print(d['parity'])
PY
