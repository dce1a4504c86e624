#!/bin/bash
# round 3, trip C: full GPU suite (parity recorded), next-segment prefetch A/B, bench in the driver's regime
O=gpurun_out/r3c; mkdir -p $O
rm -f $O/parity_record.jsonl
UMFA_PARITY_RECORD=$PWD/$O/parity_record.jsonl timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1
tail -15 $O/gpu_tests.txt
LIBS="base=tools/lab_bin/libMFAFFI_base.so nopref=tools/lab_bin/libMFAFFI_nopref.so new=intree"
AB="timeout 300 python tools/ab_inproc.py"
$AB $LIBS > $O/ab_flux.json 2>$O/ab_err.txt
$AB --causal $LIBS > $O/ab_flux_causal.json 2>>$O/ab_err.txt
$AB --shape 4,16,8192,128 --causal --rounds 6 --inner 5 $LIBS > $O/ab_causal.json 2>>$O/ab_err.txt
$AB --shape 1,16,8192,128 $LIBS > $O/ab_s8192.json 2>>$O/ab_err.txt
$AB --shape 8,16,1024,128 --causal $LIBS > $O/ab_b8s1024c.json 2>>$O/ab_err.txt
$AB --shape 1,3,4096,128 $LIBS > $O/ab_h3.json 2>>$O/ab_err.txt
$AB --quant 2 $LIBS > $O/ab_flux_i8.json 2>>$O/ab_err.txt
cat $O/ab_*.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_regime.json 2>$O/bench_err.txt
tail -c 600 $O/bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c/bench_driver_regime.json'))
print(d['value'], d['ms_per_step'], d['settle'], d['roofline']['frac'])
for k,v in d['configs'].items(): print(k, {a:b for a,b in v.items() if a in('ms','tflops','frac','kernel','rel','rms','format_floor_rel')})
print(d['int8'])
PY
