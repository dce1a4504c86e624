#!/usr/bin/env python3
"""GQA inference: zero-copy views vs repeat_interleave (python tools/bench_gqa.py B Hq Hkv S D [causal])"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, Hq, Hkv, S, D = (int(x) for x in sys.argv[1:6])
causal = len(sys.argv) > 6 and sys.argv[6] == "causal"
q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16)
g = Hq // Hkv
def t(fn):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    x = sorted(a.elapsed_time(b) for a, b in ev)
    return x[len(x) // 2] * 1e3
zc = t(lambda: umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=True))
kz = umfa_torch.last_kernel()
ex = t(lambda: umfa_torch.scaled_dot_product_attention(q, k.repeat_interleave(g, 1).contiguous(), v.repeat_interleave(g, 1).contiguous(), is_causal=causal))
print(f"B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} causal={int(causal)}: zero-copy views {zc:.1f} us [{kz}]   repeat_interleave route {ex:.1f} us [{umfa_torch.last_kernel()}]")
