#!/usr/bin/env python3
"""Decode-shaped forward (few query rows, long K/V): python tools/bench_decode.py B H Sq Skv D"""
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
for _ in range(5): umfa_torch.attention_forward(q, k, v, out=out)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); umfa_torch.attention_forward(q, k, v, out=out); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
med = t[len(t) // 2]
byts = 2 * B * H * Skv * D * 2
print(f"B{B} H{H} Sq{Sq} Skv{Skv} D{D}: {med*1e3:8.1f} us  K+V {byts/1e6:.0f} MB -> {byts/med/1e6:.0f} GB/s  [{umfa_torch.last_kernel()}]")
