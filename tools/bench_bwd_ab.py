#!/usr/bin/env python3
"""in-stream backward (bf16 gradients), per-kernel-pair time with HIP events: python tools/bench_bwd_ab.py B H S D [causal]
run with UMFA_LIBRARY=... for another build"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, S, D = (int(x) for x in sys.argv[1:5])
causal = len(sys.argv) > 5 and sys.argv[5] == "causal"
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, return_lse=True)
f = lambda: umfa_torch.attention_backward(do, q, k, v, o, lse, scale=D ** -0.5, causal=causal)
for _ in range(5):
    f()
ts = []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
fl = 10.0 * B * H * S * S * D * (0.5 if causal else 1.0)
print(f"B{B} H{H} S{S} D{D} causal={int(causal)} backward median {ts[15]:.4f} ms min {ts[0]:.4f}  {fl / ts[15] / 1e9:.1f} TFLOP/s [{umfa_torch.last_kernel()}]")
