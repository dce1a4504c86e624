#!/bin/bash
# round 4, trip AG: gate refinement: full GPU suite
O=gpurun_out/r4ag; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -6 $O/tests.txt | cut -c1-300
