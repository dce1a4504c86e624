#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2 3; do
 for lib in new prev; do
  if [ $lib = prev ]; then export UMFA_LIBRARY=tools/lab_bin/libMFAFFI_prev.so; else unset UMFA_LIBRARY; fi
  echo "== $lib"
  for a in "2 16 4096 64 bf16" "2 16 4096 64 bf16 causal" "1 16 8192 64 fp16" "4 32 2048 64 bf16"; do python tools/bench_bwd.py $a 2>&1 | grep -v amdgpu.ids; done
 done
done
