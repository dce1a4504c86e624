#!/bin/bash
# round 4, trip AS: row sums on the matrix pipe in the 128-row kernel at head_dim <= 64: stamps / timing A/B (lab binaries), then tests
O=gpurun_out/r4as; mkdir -p $O
export TMPDIR=/tmp
for v in nomsum msum nomsum msum; do timeout 60 tools/lab_bin/cfg2_$v 16 1024 50 4 > $O/cfg2_$v.txt 2>&1; echo cfg2_$v $(tail -3 $O/cfg2_$v.txt | grep -o "loop [0-9.]* us\|last end [0-9.]* us\|median [0-9.]* us" | tr '\n' ' '); done
for v in nomsum_nc msum_nc nomsum_nc msum_nc; do echo d64_$v B8H16S1024: $(timeout 60 tools/lab_bin/d64_$v 16 1024 50 8 | tail -1 | cut -c1-120); done
for v in nomsum_nc msum_nc; do echo d64_$v B2H24S4096: $(timeout 60 tools/lab_bin/d64_$v 24 4096 30 2 | tail -1 | cut -c1-120); done
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
