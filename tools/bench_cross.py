#!/usr/bin/env python3
"""Cross-attention shapes (long Sq, short Skv): python tools/bench_cross.py B H Sq Skv D"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, Sq, Skv, D = (int(x) for x in sys.argv[1:6])
q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
out = torch.empty_like(q)
def run(): umfa_torch.attention_forward(q, k, v, out=out)
for _ in range(5): run()
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): run()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
import time
t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
fl = 4.0 * B * H * Sq * Skv * D
byts = (2 * B * H * Sq * D + 2 * B * H * Skv * D) * 2
print(f"B{B} H{H} Sq{Sq} Skv{Skv} D{D}: {dt*1e6:8.1f} us/step  {fl/dt/1e12:7.1f} TFLOP/s  Q+O+K+V {byts/1e6:.0f} MB -> {byts/dt/1e9:.0f} GB/s  [{umfa_torch.last_kernel()}]")
