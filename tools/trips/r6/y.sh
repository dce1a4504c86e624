#!/bin/bash
# trip y: the whole GPU suite + smoke at the balanced-causal-pairs build
O=gpurun_out/r6y; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/gpu_suite.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
