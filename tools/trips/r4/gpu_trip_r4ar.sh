#!/bin/bash
O=gpurun_out/r4ar; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python -m pytest tests/test_gpu_smoke_entry.py -m gpu -q > $O/tests.txt 2>&1; tail -2 $O/tests.txt
