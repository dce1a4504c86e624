#!/usr/bin/env python3
"""Does a hipGraph of K forward launches close the gap between wall time per step and kernel time?"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, S, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 24, 4096, 128)
causal = len(sys.argv) > 5 and sys.argv[5] == "causal"
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
K = 50
def step(): umfa_torch.attention_forward(q, k, v, causal=causal, out=out)
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / K
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / K
fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
print(f"B{B} H{H} S{S} D{D} causal={int(causal)} [{umfa_torch.last_kernel()}] eager {eager*1e6:.1f} us/step ({fl/eager/1e12:.0f} TF)   graph {graph*1e6:.1f} us/step ({fl/graph/1e12:.0f} TF)")
