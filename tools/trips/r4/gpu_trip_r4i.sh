#!/bin/bash
# round 4, trip I: bool mask tensors on the one-wave-per-SIMD kernels: parity first, then the whole suite, masked FLUX timings
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_w64_masks.py -x -q > $O/tests_masks.txt 2>&1; tail -30 $O/tests_masks.txt | cut -c1-300
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -6 $O/tests.txt | cut -c1-250
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4i/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('cfg3_flux_bf16_mask_padding','cfg3_flux_bf16_mask_blockdiag'): print(k, d['configs'][k])
PY
tail -3 $O/bench_err.txt
