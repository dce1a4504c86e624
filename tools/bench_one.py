#!/usr/bin/env python3
"""python tools/bench_one.py B H S D [causal] -- kernel-only median time of the bf16 forward"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, S, D = (int(x) for x in sys.argv[1:5])
causal = len(sys.argv) > 5 and sys.argv[5] == "causal"
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
for _ in range(10):
    umfa_torch.attention_forward(q, k, v, causal=causal, out=out)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); umfa_torch.attention_forward(q, k, v, causal=causal, out=out); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
med = t[len(t) // 2]
fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
print(f"B{B} H{H} S{S} D{D} causal={int(causal)} {med*1e3:8.1f} us (min {t[0]*1e3:.1f}) {fl/med/1e9:8.1f} TFLOP/s")
