#!/bin/bash
# trip o: fuzz of the round's new mask kernels (2000 seeds of run_w64_mask_case) + 300 seeds of every leg on the round's build
O=gpurun_out/r6o; mkdir -p $O
timeout 1500 python3 tools/lab/value_fuzz.py 1000 2000 run_w64_mask_case 2>&1 | grep -v amdgpu > $O/fuzz_w64_mask.txt; tail -6 $O/fuzz_w64_mask.txt
timeout 1500 python3 tools/lab/value_fuzz.py 70000 300 2>&1 | grep -v amdgpu > $O/soak.txt; tail -6 $O/soak.txt
