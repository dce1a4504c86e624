#!/bin/bash
# round 4, trip N: key-split form of the head_dim-64 kernel: stamps binaries (ring depth 2 / 3 / 4), probe (A/B + parity), the D = 64 tests
O=gpurun_out/r4n; mkdir -p $O
export TMPDIR=/tmp
for v in ks1_ns2 ks2_ns2 ks2_ns3 ks2_ns4; do timeout 120 tools/lab_bin/cfg2_lab_$v 16 1024 50 4 > $O/stamps_$v.txt 2>&1; echo $v; tail -3 $O/stamps_$v.txt | cut -c1-300; done
timeout 900 python tools/lab/ksplit_probe.py > $O/ksplit_probe.json 2> $O/ksplit_probe_err.txt; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4n/ksplit_probe.json'))
for k,v in d.items(): print(k, v['ks2_us'], v['four_wave_us'], v['speedup'], v.get('ks2_rel'), v['ks2_vs_four_wave_rel'])
PY
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_forward.py tests/test_gpu_w64.py tests/test_gpu_value_fuzz.py tests/test_gpu_fuzz.py tests/test_gpu_backward.py -m gpu -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-250
