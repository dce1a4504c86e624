#!/usr/bin/env python3
"""debug: the additive-mask kernel against the 128-row kernel on synthetic masks that isolate the mask path's addressing"""
import sys
sys.path[:0] = [".", "universal-metal-flash-attention_amd"]
import torch
import umfa_torch

torch.manual_seed(0)
B, H, S, D = 1, 1, 256, 128
Skv = int(sys.argv[1]) if len(sys.argv) > 1 else 128
q = torch.randn(B, H, S, D, device="cuda", dtype=torch.float16)
k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.float16)
v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.float16)
i = torch.arange(S, device="cuda")[:, None].float()
j = torch.arange(Skv, device="cuda")[None, :].float()
masks = {
    "const -1.5": torch.full((S, Skv), -1.5, device="cuda"),
    "row only": -(i % 7) * torch.ones(1, Skv, device="cuda"),
    "key only (materialised)": -(j % 5) * torch.ones(S, 1, device="cuda"),
    "key only (row stride 0)": (-(j % 5)).expand(S, Skv),
    "one key -inf": torch.where(j == 3, float("-inf"), 0.0) * torch.ones(S, 1, device="cuda"),
    "key j masked for row i==j": torch.where(i == j, float("-inf"), 0.0),
    "random": torch.randn(S, Skv, device="cuda") * 2,
}
with umfa_torch.options(force_w64=1):
    for name, m in masks.items():
        m16 = m.to(torch.float16)
        m16 = m16[None, None] if m16.is_contiguous() else m16[None, None]
        o = umfa_torch.attention_forward(q, k, v, mask=m16, out_dtype=torch.float32)
        kn = umfa_torch.last_kernel()
        with umfa_torch.options(no_w64_bias=1, force_w64=0):
            r = umfa_torch.attention_forward(q, k, v, mask=m16, out_dtype=torch.float32)
            kr = umfa_torch.last_kernel()
        err = (o - r).abs().amax(-1)[0, 0]  # per row
        bad = (err > 1e-2 * r.abs().max()).nonzero().flatten().tolist()
        print(f"{name:32s} {kn} vs {kr}: max {float(err.max()):.3e} of {float(r.abs().max()):.3f}; bad rows {len(bad)}: {bad[:24]}")
