#!/usr/bin/env python3
"""Launch the FLUX-shape forward with an additive fp16 relative-position bias N times (profiling driver for rocprofv3)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H, S, D = 1, 24, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
i = torch.arange(S, device="cuda")
m = (-(i[:, None] - i[None, :]).abs().to(torch.float16) / 256.0)[None, None].contiguous()
for _ in range(n):
    umfa_torch.attention_forward(q, k, v, mask=m, out=out)
torch.cuda.synchronize()
print("done", umfa_torch.last_kernel())
