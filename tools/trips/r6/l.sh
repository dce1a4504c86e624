#!/bin/bash
# trip l: large additive masks without the classification pass + the additive-mask suite + timing
O=gpurun_out/r6l; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_bias.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python3 tools/lab/bias_probe.py 2>&1 | grep -v amdgpu | grep "per-head\|rel_pos" | tee $O/bias_probe.jsonl
