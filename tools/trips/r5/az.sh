#!/bin/bash
# trip az: long soak of the value fuzz's quantised legs (the ones whose value ranges the round widened) + masks + graphs
O=gpurun_out/r5az; mkdir -p $O
for leg in run_i8_case run_qbwd_case run_prequant_case run_mask_case run_graph_case run_streams_case; do
  timeout 900 python3 tools/lab/value_fuzz.py 20000 4000 $leg 2>&1 | grep -v amdgpu | tail -3 | sed "s/^/$leg: /" | tee -a $O/soak.txt
done
