#!/usr/bin/env python3
"""Lab: umfa_quantized_backward_stream at config 4's shape, n calls (for a kernel trace).  python tools/lab/qbwd_stream_loop.py [n]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
B, H, S, D = 1, 16, 8192, 128
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, return_lse=True)
for _ in range(n):
    umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse)
torch.cuda.synchronize()
print(umfa_torch.last_kernel())
