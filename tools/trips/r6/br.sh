#!/bin/bash
# trip br: the size rule of the mask pre-passes for masks without a head dimension (lab option mask_pass_ratio): rule 2 against 8, fp16 / fp32, both kernels
O=gpurun_out/r6br; mkdir -p $O
timeout 1500 python3 tools/lab/mask_pass_rule_probe.py $O/mask_pass_rule_probe.jsonl 2>&1 | cut -c1-700 | tail -40
