#!/usr/bin/env python3
"""Lab: the runtime-quantised forward when V is far from 1 -- O is linear in V, so O(V * s) / s should not depend on s.
python tools/lab/qfwd_range_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

torch.manual_seed(3)
for (B, H, S, D) in ((1, 8, 2048, 128), (1, 4, 512, 64)):
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    for mode in ("blockwise", "tensor", "blockwise_fp8pv"):
        base = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode=mode).double()
        kern = umfa_torch.last_kernel()
        for s in (1e-8, 1e-6, 1e-4, 1e-2, 1e2, 1e4, 1e6):
            # powers of two would be exact; take them so that any difference is the fp16 image's range, not the quantiser's rounding
            s2 = 2.0 ** round(torch.log2(torch.tensor(s)).item())
            vs = (v.float() * s2).to(torch.bfloat16)
            o = umfa_torch.quantized_attention_forward_stream(q, k, vs, quant_mode=mode).double() / s2
            torch.cuda.synchronize()
            err = float(((o - base).abs().max() / base.abs().max()).item())
            print(f"S {S} D {D} mode {mode} {kern}: V x 2^{round(torch.log2(torch.tensor(s)).item()):+d} ({s:.0e}): rel diff vs V x 1: {err:.2e} finite {bool(torch.isfinite(o).all())}")
