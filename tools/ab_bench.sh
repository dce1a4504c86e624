#!/bin/bash
# A/B on one box: bench.py (graph mode, FLUX) alternating the in-tree library and tools/lab_bin/libMFAFFI_prev.so
for i in 1 2 3; do
  for lib in new prev; do
    if [ $lib = prev ]; then export UMFA_LIBRARY=tools/lab_bin/libMFAFFI_prev.so; else unset UMFA_LIBRARY; fi
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['kernel_ms_min'])"
  done
done
