#!/bin/bash
# trip by: additive masks of any length and row alignment (the pass reads unaligned rows element by element) -- the whole GPU suite, timing, mask fuzz legs
O=gpurun_out/r6by; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/gpu_suite.txt
timeout 600 python3 tools/lab/ragged_mask_probe.py $O/ragged_mask_probe.jsonl 2>&1 | cut -c1-330 | tail -32
(time timeout 1200 python3 tools/lab/value_fuzz.py 140000 3000 run_w64_mask_case) 2>&1 | tail -9 | tee $O/fuzz_w64_mask_leg_3000_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 140000 2000 run_mask_case) 2>&1 | tail -9 | tee $O/fuzz_mask_leg_2000_seeds.txt
