#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sdpa.py tests/test_gpu_configs.py tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
 for lib in new prev; do
  if [ $lib = prev ]; then export UMFA_LIBRARY=tools/lab_bin/libMFAFFI_prev.so; else unset UMFA_LIBRARY; fi
  echo "== $lib"
  python tools/bench_bwd.py 1 24 4096 128 bf16
  python tools/bench_bwd.py 1 24 4096 128 bf16 causal
  python tools/bench_bwd.py 2 16 4096 64 bf16
  python tools/bench_bwd.py 1 16 8192 128 fp16
 done
done
