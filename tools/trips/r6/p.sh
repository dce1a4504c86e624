#!/bin/bash
# trip p: soak of all nineteen legs on the round's build
O=gpurun_out/r6p; mkdir -p $O
timeout 3000 python3 tools/lab/value_fuzz.py 80000 2500 2>&1 | grep -v amdgpu > $O/soak.txt; tail -8 $O/soak.txt
