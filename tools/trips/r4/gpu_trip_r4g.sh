#!/bin/bash
# round 4, trip G: lazy rebase threshold of fp16 P (2^6 vs 2^10) A/B; HBM traffic of the headline call (cast pass + attention
# kernel, separate --pmc passes); SQ counters of the default headline kernel; bench with the masked FLUX entries
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --parity hi6=tools/lab_bin/libMFAFFI_hi6.so hi10=tools/lab_bin/libMFAFFI_hi10.so > $O/ab_lazy_hi_flux.json 2>$O/ab_err.txt; cat $O/ab_lazy_hi_flux.json
timeout 600 python tools/ab_inproc.py --shape 1,16,8192,128 --out fp32 hi6=tools/lab_bin/libMFAFFI_hi6.so hi10=tools/lab_bin/libMFAFFI_hi10.so > $O/ab_lazy_hi_s8192.json 2>>$O/ab_err.txt; cat $O/ab_lazy_hi_s8192.json
timeout 600 python tools/ab_inproc.py --shape 4,16,4096,128 --causal --out fp32 hi6=tools/lab_bin/libMFAFFI_hi6.so hi10=tools/lab_bin/libMFAFFI_hi10.so > $O/ab_lazy_hi_causal.json 2>>$O/ab_err.txt; cat $O/ab_lazy_hi_causal.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/run_fwd.py 10 > /dev/null 2>$O/prof_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/run_fwd.py 10 > /dev/null 2>>$O/prof_err.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 tools/run_fwd.py 10 > /dev/null 2>>$O/prof_err.txt
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
find $O -name "*.db" -delete; rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench_err.txt; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4g/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k:(v.get('ms'),v.get('frac'),v.get('rel')) for k,v in d['configs'].items()})
PY
tail -3 $O/bench_err.txt
