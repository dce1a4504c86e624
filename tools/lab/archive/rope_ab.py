"""Fused RoPE + SDPA entry against the unfused sequence (rotate q, rotate k, attend): HIP events on torch's stream,
interleaved rounds, medians.  python tools/lab/rope_ab.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "universal-metal-flash-attention_amd"))
import torch
import umfa_torch
from umfa_torch import ops

def tables(S, D):
    ang = torch.rand(S, D // 2) * 6.283
    return ang.cos().repeat_interleave(2, -1).cuda(), ang.sin().repeat_interleave(2, -1).cuda()

def med(f, n=30):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

for (B, H, S, D, causal) in [(1, 24, 4096, 128, False), (1, 16, 8192, 128, False), (4, 16, 8192, 128, True), (2, 24, 2048, 128, False)]:
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    cos, sin = tables(S, D)
    fused = lambda: ops.rope_attention_forward(q, k, v, cos, sin, causal=causal)
    unfused = lambda: ops.attention_forward(ops.rope_rotate(q, cos, sin), ops.rope_rotate(k, cos, sin), v, causal=causal)
    plain = lambda: ops.attention_forward(q, k, v, causal=causal)
    for f in (fused, unfused, plain):
        for _ in range(5): f()
    torch.cuda.synchronize()
    r = {"fused": [], "unfused": [], "plain": []}
    for _ in range(3):
        r["fused"].append(med(fused)); r["unfused"].append(med(unfused)); r["plain"].append(med(plain))
    print(f"B{B} H{H} S{S} causal{int(causal)}: " + "  ".join(f"{k_} {min(v_):.4f} ms" for k_, v_ in r.items()), umfa_torch.last_kernel(), flush=True)
