#!/bin/bash
# trip bb: the two new fuzz legs -- their test seeds, then a soak
O=gpurun_out/r5bb; mkdir -p $O
python3 -m pytest tests/test_gpu_value_fuzz.py -m gpu -x -q -k "wide_head or caller_masks" 2>&1 | tail -4
for leg in run_wide_case run_qmask_case; do
  timeout 1500 python3 tools/lab/value_fuzz.py 40000 1500 $leg 2>&1 | grep -v amdgpu | tail -6 | sed "s/^/$leg: /" | tee -a $O/soak.txt
done
