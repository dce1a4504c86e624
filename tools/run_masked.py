#!/usr/bin/env python3
"""Profiling driver: FLUX-shape forward with a mask tensor / window (python tools/run_masked.py n kind [dtype]);
kind: blockdiag | padding | window_tensor | window | causal | random | additive_blockdiag | bias_f32 (fp16 holds it) | bias_f32_inexact"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "blockdiag"
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
B, H, S, D = 1, 24, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
i = torch.arange(S, device="cuda")
kw = {}
if kind == "blockdiag":
    kw["mask"] = ((i[:, None] // 1024) == (i[None, :] // 1024))[None, None].contiguous()
elif kind == "additive_blockdiag":
    m = ((i[:, None] // 1024) == (i[None, :] // 1024))[None, None]
    kw["mask"] = torch.where(m, 0.0, float("-inf")).to(torch.float32).contiguous()
elif kind == "bias_f32":
    kw["mask"] = (-(i[:, None] - i[None, :]).abs().float() / 256.0).to(torch.float16).float()[None, None].contiguous()
elif kind == "bias_f32_inexact":
    kw["mask"] = (-(i[:, None] - i[None, :]).abs().float() / 256.0)[None, None].contiguous()
elif kind == "padding":
    kw["mask"] = (i < 3000)[None, None, None, :].contiguous()
elif kind == "window_tensor":
    kw["mask"] = ((i[:, None] - i[None, :]).abs() <= 512)[None, None].contiguous()
elif kind == "window":
    kw["window"] = (512, 512)
elif kind == "causal":
    kw["causal"] = True
elif kind == "random":
    kw["mask"] = (torch.rand(1, H, S, S, device="cuda") > 0.5)
for _ in range(n):
    umfa_torch.attention_forward(q, k, v, out=out, **kw)
torch.cuda.synchronize()
print(kind, umfa_torch.last_kernel())
