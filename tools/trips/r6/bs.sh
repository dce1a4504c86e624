#!/bin/bash
# trip bs: the shape-aware size rule of the mask pre-passes (8 x for [B,1,Sq,Skv] float masks) -- the probe again with the default rule, the whole GPU suite, the mask and
# forward fuzz legs
O=gpurun_out/r6bs; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_suite.txt
timeout 600 python3 - <<'PY' 2>&1 | tail -12 | tee $O/default_rule_check.txt
import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/universal-metal-flash-attention_amd', '/root/repo/tools']
import torch, umfa_torch
from bench_mask_f32 import graph_us
NEG = float("-inf")
for (B, H, S, D) in [(8, 2, 4096, 128), (4, 4, 4096, 64)]:
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    i = torch.arange(S, device="cuda")
    docs = torch.where((i[:, None] // (S // 4)) == (i[None, :] // (S // 4)), 0.0, NEG)
    for mdt in (torch.float16, torch.bfloat16, torch.float32):
        m = docs.to(mdt)[None, None].expand(B, 1, S, S).contiguous()
        t = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
        kern = umfa_torch.last_kernel()
        with umfa_torch.options(mask_pass_ratio=2, f32_mask_ratio=2):
            t2 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
        print(json.dumps({"shape": f"B{B} H{H} S{S} D{D}", "mask": f"documents [B,1,S,S] {mdt}", "default_rule_us": round(t, 1), "rule_2_us": round(t2, 1), "kernel": kern.split(" (")[0]}))
PY
(time timeout 1200 python3 tools/lab/value_fuzz.py 80000 1500 run_w64_mask_case) 2>&1 | tail -4 | tee $O/fuzz_w64_mask_leg_1500_seeds.txt
(time timeout 1200 python3 tools/lab/value_fuzz.py 80000 1500 run_mask_case) 2>&1 | tail -4 | tee $O/fuzz_mask_leg_1500_seeds.txt
