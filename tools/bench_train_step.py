#!/usr/bin/env python3
"""fwd + bwd of the FLUX shape through the SDPA autograd path, wall time per iteration"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, S, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 24, 4096, 128)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16)
def it():
    o = umfa_torch.scaled_dot_product_attention(q, k, v)
    o.backward(do)
    q.grad = k.grad = v.grad = None
for _ in range(5): it()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n): it()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 14.0 * B * H * S * S * D  # 4 forward + 10 backward
print(f"B{B} H{H} S{S} D{D} fwd+bwd {dt*1e3:.3f} ms/iter  {fl/dt/1e12:.0f} TFLOP/s algorithmic")
