#!/bin/bash
# trip bu: finfo.min masks (the transformers idiom) -- -inf on every route, their tiles skipped; tests, probe, mask fuzz legs, the whole suite
O=gpurun_out/r6bu; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_suite.txt
timeout 600 python3 tools/lab/hf_mask_probe.py $O/hf_mask_probe.jsonl 2>&1 | cut -c1-420 | tail -20
(time timeout 1200 python3 tools/lab/value_fuzz.py 90000 2000 run_w64_mask_case) 2>&1 | tail -9 | tee $O/fuzz_w64_mask_leg_2000_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 90000 2000 run_mask_case) 2>&1 | tail -9 | tee $O/fuzz_mask_leg_2000_seeds.txt
