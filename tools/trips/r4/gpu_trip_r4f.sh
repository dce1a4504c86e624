#!/bin/bash
# round 4, trip F: deferred reference in the 128-row kernel (config 2), quantiser with packed inputs / 8 workgroups per CU: parity, timing
O=gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -8 $O/tests.txt | cut -c1-250
for pv in 0 1; do timeout 120 tools/lab_bin/cfg2_lab_def_pv$pv 16 1024 50 4 > $O/cfg2_stamps_def_pv$pv.txt 2>&1; tail -3 $O/cfg2_stamps_def_pv$pv.txt; done
timeout 600 python tools/bench_cfg2.py > $O/bench_cfg2.json 2>$O/cfg2_err.txt; python3 -c "
import json
d=json.loads(open('$O/bench_cfg2.json').read())
for k,v in d.items(): print(k, v['nosplit'], v['nosplit_kernel'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pair -- python3 tools/run_pair.py 20 1 24 4096 128 > $O/pair.txt 2>$O/prof_err.txt
find $O/pair -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/int8_vs_bf16_flux_kernel_stats.csv; cut -c1-160 $O/int8_vs_bf16_flux_kernel_stats.csv | head -5
rm -rf $O/pair
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['int8'].items(): print(k, v['bf16_ms'], v['int8_ms_incl_quantiser'], v['speedup'], v['fp8pv_ms_incl_quantiser'], v['fp8pv_speedup'])
print({k:(v.get('ms'),v.get('frac')) for k,v in d['configs'].items()})
PY
find $O -name "*.db" -delete
