#!/usr/bin/env python3
"""a few in-stream backward launches of one shape (profiling target): python tools/run_bwd.py B H S D [n]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, S, D = (int(x) for x in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 5
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
o, lse = umfa_torch.attention_forward(q, k, v, return_lse=True)
for _ in range(n):
    umfa_torch.attention_backward(do, q, k, v, o, lse, scale=D ** -0.5)
torch.cuda.synchronize()
