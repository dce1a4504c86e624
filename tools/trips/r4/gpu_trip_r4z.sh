#!/bin/bash
# round 4, trip Z: mask schedule fuzz (new), twice
O=gpurun_out/r4z; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_w64_masks.py -m gpu -q -k "fuzz or few_blocks" > $O/tests_$i.txt 2>&1; tail -4 $O/tests_$i.txt | cut -c1-400; done
grep -n "^E  " $O/tests_1.txt | head -10 | cut -c1-300
