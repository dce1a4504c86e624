#!/bin/bash
# bench.py (graph mode) for the in-tree library and every tools/lab_bin variant, two passes, same box
for pass in 1 2; do
  for lib in "" tools/lab_bin/libMFAFFI_*.so; do
    if [ -n "$lib" ]; then export UMFA_LIBRARY=$lib; else unset UMFA_LIBRARY; fi
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); i=d['int8']; print('%-40s' % '${lib:-in-tree}', d['value'], d['roofline']['kernel_ms_mean'], 'int8 flux', i['flux_B1_H24_S4096_D128']['int8_ms_incl_quantiser'], 'cfg4', i['cfg4_B1_H16_S8192_D128']['int8_ms_incl_quantiser'], 'bf16 cfg4', i['cfg4_B1_H16_S8192_D128']['bf16_ms'])"
  done
done
