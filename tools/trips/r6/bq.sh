#!/bin/bash
# trip bq: the whole GPU suite at the last commit of the round (Tuning option f32_mask_ratio, the backward leg's second reference + its regression test), smoke
O=gpurun_out/r6bq; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_suite.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
