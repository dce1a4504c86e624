#!/usr/bin/env python3
"""Launch the bf16 FLUX-shape forward N times (profiling driver for rocprofv3)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H, S, D = (int(x) for x in sys.argv[2:6]) if len(sys.argv) >= 6 else (1, 24, 4096, 128)
causal = len(sys.argv) > 6 and sys.argv[6] == "causal"
odt = torch.bfloat16 if (len(sys.argv) > 7 and sys.argv[7] == "bf16o") or (len(sys.argv) > 6 and sys.argv[6] == "causal") else torch.float32  # the headline writes fp32 O
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty(B, H, S, D, device="cuda", dtype=odt)
for _ in range(n):
    umfa_torch.attention_forward(q, k, v, causal=causal, out=out)
torch.cuda.synchronize()
print("done", umfa_torch.last_kernel())
