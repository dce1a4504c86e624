#!/usr/bin/env python3
"""Kernel-only timing (HIP events) of the forward over several shapes: python tools/bench_shapes.py [dtype]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

shapes = [(1, 24, 4096, 128, False), (2, 16, 4096, 128, False), (4, 24, 4096, 128, False), (1, 16, 8192, 128, False),
          (4, 16, 1024, 64, True), (1, 24, 4096, 128, True), (1, 32, 4096, 64, False), (16, 16, 2048, 128, False)]
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
for B, H, S, D, causal in shapes:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
    out = torch.empty_like(q)
    for _ in range(5):
        umfa_torch.attention_forward(q, k, v, causal=causal, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); umfa_torch.attention_forward(q, k, v, causal=causal, out=out); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    med = t[len(t) // 2]
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    nwg = B * H * ((S + 127) // 128)
    print(f"B{B} H{H} S{S} D{D} causal={int(causal)} wgs={nwg:5d} ({nwg/512:.2f} rounds)  {med*1e3:8.1f} us  {fl/med/1e9:8.1f} TFLOP/s  [{umfa_torch.last_kernel()}]")
