#!/bin/bash
# round 4, trip P: software-pipelined loop of the head_dim-64 kernel: stamps, A/B probe of the three forms, tests
O=gpurun_out/r4p; mkdir -p $O
export TMPDIR=/tmp
for v in lab_ks1_ns2 pipe; do timeout 120 tools/lab_bin/cfg2_$v 16 1024 50 4 > $O/stamps_$v.txt 2>&1; echo $v; tail -3 $O/stamps_$v.txt | cut -c1-300; done
timeout 900 python tools/lab/ksplit_probe.py > $O/forms_probe.json 2> $O/forms_probe_err.txt; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4p/forms_probe.json'))
for k,v in d.items(): print(k, v['pipe_us'], v['ks2_us'], v['four_wave_us'], v['speedup_pipe'], v.get('pipe_rel'), v['pipe_vs_four_wave_rel'], v['pipe_lse_max_abs_diff'], v['finite'])
PY
tail -3 $O/forms_probe_err.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_forward.py tests/test_gpu_w64.py tests/test_gpu_value_fuzz.py tests/test_gpu_fuzz.py tests/test_gpu_backward.py -m gpu -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-250
