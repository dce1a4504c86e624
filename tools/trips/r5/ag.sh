#!/bin/bash
# trip ag: soak of the value fuzz (every leg) on the round's last build
O=gpurun_out/r5ag; mkdir -p $O
timeout 2400 python3 tools/lab/value_fuzz.py 10000 3000 > $O/fuzz.txt 2>&1; tail -8 $O/fuzz.txt | cut -c1-400
