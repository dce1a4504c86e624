#!/bin/bash
# trip bn: the size rule of the fp32-mask route -- pair against 128-row kernel alone as the mask grows against the call's tensors (lab option f32_mask_ratio)
O=gpurun_out/r6bn; mkdir -p $O
timeout 900 python3 tools/lab/f32_mask_ratio_probe.py $O/f32_mask_ratio_probe.jsonl 2>&1 | cut -c1-260 | tail -40
timeout 600 python3 -m pytest tests/test_gpu_w64_f32_mask.py tests/test_gpu_w64_bias.py -q 2>&1 | tail -3
