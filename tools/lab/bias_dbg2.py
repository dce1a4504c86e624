#!/usr/bin/env python3
"""debug: read back the EFFECTIVE additive mask the bias kernel applied (V = identity -> O = P), for key-coded and row-coded masks in tile 1"""
import sys
sys.path[:0] = [".", "universal-metal-flash-attention_amd"]
import torch
import umfa_torch

torch.manual_seed(0)
B, H, S, D, Skv = 1, 1, 256, 128, 128
q = (torch.randn(B, H, S, D, device="cuda") * 0.05).to(torch.float16)
k = (torch.randn(B, H, Skv, D, device="cuda") * 0.05).to(torch.float16)
v = torch.eye(Skv, D, device="cuda", dtype=torch.float16)[None, None].contiguous()
i = torch.arange(S, device="cuda")[:, None]
j = torch.arange(Skv, device="cuda")[None, :]
with umfa_torch.options(force_w64=1):
    ou = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    for name, m in (("key-coded", torch.where(j >= 64, -(j - 64).float() / 4, 0.0).expand(S, Skv).contiguous()),
                    ("row-coded", torch.where(j >= 64, -(i % 64).float() / 4, 0.0).expand(S, Skv).contiguous())):
        m16 = m.to(torch.float16)[None, None].contiguous()
        om = umfa_torch.attention_forward(q, k, v, mask=m16, out_dtype=torch.float32)
        print(name, umfa_torch.last_kernel())
        eff = torch.log(om / ou)[0, 0]
        eff = eff - eff[:, :1]
        code = (-eff * 4).round()
        for r in (0, 1, 2, 8, 9, 31, 32, 33, 63, 64, 65, 130, 255):
            print(f" row {r:3d} tile0[:8] {code[r, :8].int().tolist()} tile1: {code[r, 64:].int().tolist()}")
