#!/usr/bin/env python3
"""Lab: FLUX-shape forward + backward through the torch SDPA surface (autograd), n steps -- for a kernel trace: which kernels does a training step launch besides ours?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
q, k, v = (torch.randn(1, 24, 4096, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
w = torch.randn(1, 24, 4096, 128, device="cuda", dtype=torch.bfloat16)
for _ in range(n):
    o = umfa_torch.scaled_dot_product_attention(q, k, v)
    o.backward(w)
    q.grad = k.grad = v.grad = None
torch.cuda.synchronize()
print(umfa_torch.last_kernel())
