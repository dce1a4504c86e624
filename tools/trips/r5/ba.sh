#!/bin/bash
# trip ba: long soak of the other legs of the value fuzz
O=gpurun_out/r5ba; mkdir -p $O
for leg in run_case run_bwd_case run_shape_case run_gqa_case run_rope_case run_host_case run_aux_case run_threads_case run_big_case run_bwd_shape_case; do
  timeout 1200 python3 tools/lab/value_fuzz.py 30000 2500 $leg 2>&1 | grep -v amdgpu | tail -3 | sed "s/^/$leg: /" | tee -a $O/soak.txt
done
