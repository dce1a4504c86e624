#!/bin/bash
# trip bz: the 128-row kernel reads a realigned copy of masks with unaligned rows -- the whole GPU suite, timing, the mask fuzz legs
O=gpurun_out/r6bz; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/gpu_suite.txt
timeout 600 python3 tools/lab/mask_realign_probe.py $O/mask_realign_probe.jsonl 2>&1 | cut -c1-300 | tail -24
(time timeout 1200 python3 tools/lab/value_fuzz.py 150000 3000 run_mask_case) 2>&1 | tail -9 | tee $O/fuzz_mask_leg_3000_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 150000 2000 run_w64_mask_case) 2>&1 | tail -9 | tee $O/fuzz_w64_mask_leg_2000_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 150000 1000 run_qmask_case) 2>&1 | tail -9 | tee $O/fuzz_qmask_leg_1000_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 150000 1000 run_graph_case) 2>&1 | tail -9 | tee $O/fuzz_graph_leg_1000_seeds.txt
