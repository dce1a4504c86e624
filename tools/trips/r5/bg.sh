#!/bin/bash
# trip bg: soak of all eighteen legs on the round's last build
O=gpurun_out/r5bg; mkdir -p $O
timeout 3300 python3 tools/lab/value_fuzz.py 60000 2500 2>&1 | grep -v amdgpu | tail -8 | tee $O/soak.txt
