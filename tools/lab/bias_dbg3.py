#!/usr/bin/env python3
"""debug: which 64-row blocks of which head go wrong with a forced small grid (cut blocks, several segments per workgroup)"""
import sys
sys.path[:0] = [".", "universal-metal-flash-attention_amd"]
import torch
import umfa_torch

torch.manual_seed(0)
B, H, Sq, Skv, D = 1, 3, 1280, 1408, 128
ODT = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
i = torch.arange(Sq, device="cuda")[:, None]
j = torch.arange(Skv, device="cuda")[None, :]
masks = {"rel_pos": (-(i - j).abs().float() / 64.0).to(torch.float16)[None, None].contiguous(),
         "all_zero": torch.zeros(1, 1, Sq, Skv, device="cuda", dtype=torch.float16),
         "const": torch.full((1, 1, Sq, Skv), -1.0, device="cuda", dtype=torch.float16)}
for name, m in masks.items():
    with umfa_torch.options(no_w64_bias=1):
        r = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=ODT)
    with umfa_torch.options(force_w64=1, w64_grid=grid):
        o = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=ODT)
        kn = umfa_torch.last_kernel()
    err = (o.float() - r.float()).abs().amax(-1)[0]  # [H, Sq]
    blk = err.view(H, Sq // 64, 64).amax(-1)
    print(name, kn, "max", float(err.max()))
    for h in range(H):
        print("  head", h, ["%.0e" % float(x) for x in blk[h]])
