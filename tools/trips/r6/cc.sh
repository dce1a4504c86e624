#!/bin/bash
# trip cc: the guarded pair's second route reads a realigned copy of an fp32 mask with unaligned rows -- mask suites, timing of the odd-length fp32 case, mask fuzz leg
O=gpurun_out/r6cc; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_f32_mask.py tests/test_gpu_w64_bias.py tests/test_gpu_w64_masks.py tests/test_gpu_forward.py -q 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python3 - <<'PY' 2>&1 | tail -8 | tee $O/fp32_odd_length.txt
import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/universal-metal-flash-attention_amd', '/root/repo/tools']
import torch, umfa_torch
from bench_mask_f32 import graph_us
for (B, H, S, D) in [(1, 24, 4097, 128), (2, 16, 3001, 64), (4, 16, 1111, 128)]:
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    i = torch.arange(S, device="cuda")
    m = (-(i[:, None] - i[None, :]).abs().float() / 255.0)[None, None].contiguous()  # fp16 does not hold it
    t = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
    kern = umfa_torch.last_kernel().split(" (")[0]
    with umfa_torch.options(no_mask_realign=1):
        t2 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
    print(json.dumps({"shape": f"B{B} H{H} S{S} D{D}", "mask": "fp32 bias [1,1,S,S] fp16 does not hold, rows unaligned", "us": round(t, 1), "without_the_realigned_copy_us": round(t2, 1), "kernel": kern}))
PY
(time timeout 1200 python3 tools/lab/value_fuzz.py 160000 2500 run_w64_mask_case) 2>&1 | tail -9 | tee $O/fuzz_w64_mask_leg_2500_seeds.txt
