#!/bin/bash
# round 4, trip H: lazy rebase threshold of fp16 P (2^6 vs 2^10) A/B; in-kernel clock stamps of the default headline kernel and of the
# bf16 P V kernel (the guide's item 6: shader clock / real-time clock inside the kernel); HBM traffic of the headline call
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
for sh in "1,24,4096,128" "1,16,8192,128"; do timeout 600 python tools/ab_inproc.py --shape $sh --out fp32 --parity hi6=tools/lab_bin/libMFAFFI_hi6.so hi10=tools/lab_bin/libMFAFFI_hi10.so >> $O/ab_lazy_hi.jsonl 2>>$O/ab_err.txt; done
timeout 600 python tools/ab_inproc.py --shape 4,16,4096,128 --causal --out fp32 --parity hi6=tools/lab_bin/libMFAFFI_hi6.so hi10=tools/lab_bin/libMFAFFI_hi10.so >> $O/ab_lazy_hi.jsonl 2>>$O/ab_err.txt
cat $O/ab_lazy_hi.jsonl; tail -2 $O/ab_err.txt
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_stamps.so timeout 300 python tools/w64_stamps.py 1 24 4096 128 > $O/stamps_flux_default_pv16.txt 2>&1; cat $O/stamps_flux_default_pv16.txt
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_stamps.so UMFA_PV_FP16=0 timeout 300 python tools/w64_stamps.py 1 24 4096 128 > $O/stamps_flux_bf16_pv.txt 2>&1; cat $O/stamps_flux_bf16_pv.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/run_fwd.py 10 > /dev/null 2>$O/prof_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/run_fwd.py 10 > /dev/null 2>>$O/prof_err.txt
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt
find $O -name "*.db" -delete; rm -rf $O/pmc_fetch $O/pmc_write
