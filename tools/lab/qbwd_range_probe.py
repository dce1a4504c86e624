#!/usr/bin/env python3
"""Lab: the quantised backward's fp16 engine against its fp32-exact engine (option bwd_exact) when dO is far from 1 --
gradients are linear in dO, so g(dO * s) / s should not depend on s.   python tools/lab/qbwd_range_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

torch.manual_seed(3)
B, H, S, D = 1, 4, 1024, 128
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, return_lse=True)
with umfa_torch.options(bwd_exact=1):
    ref = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse)[:3]
    kern_ref = umfa_torch.last_kernel()
for s in (1.0, 1e-3, 1e-5, 1e-7, 1e-9, 1e3, 1e5, 1e8):
    dos = (do.float() * s).to(torch.bfloat16)
    with umfa_torch.options(bwd_exact=1):
        ex = umfa_torch.quantized_attention_backward_stream(dos, q, k, v, o, lse)[:3]
    g = umfa_torch.quantized_attention_backward_stream(dos, q, k, v, o, lse)
    torch.cuda.synchronize()
    errs = [float(((a.double() - b.double()).abs().max() / b.double().abs().max()).item()) for a, b in zip(g[:3], ex)]
    print(f"dO x {s:7.0e}: status {int(g[3].item())} kernel {umfa_torch.last_kernel()} rel-err vs exact engine dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}")
