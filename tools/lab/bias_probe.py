#!/usr/bin/env python3
"""FLUX-shape forward with an additive fp16 mask: the one-wave-per-SIMD bias kernel against the 128-row kernel (option no_w64_bias), graph-replayed ms."""
import json
import sys
sys.path[:0] = [".", "universal-metal-flash-attention_amd"]
import torch
import umfa_torch

dev = "cuda"


def graph_ms(fn, n=20, reps=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        g.replay()
        side.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        b.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return a.elapsed_time(b) / (n * reps)


shapes = [(1, 24, 4096, 128), (1, 16, 8192, 128), (4, 16, 2048, 128), (2, 16, 4096, 64), (4, 16, 2048, 64)]
for B, H, S, D in shapes:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
    out = torch.empty(B, H, S, D, device=dev, dtype=torch.float32)
    i = torch.arange(S, device=dev)
    masks = {"rel_pos [1,1,S,S]": (-(i[:, None] - i[None, :]).abs().to(torch.float16) / 256.0)[None, None].contiguous(),
             "blockdiag 0/-inf [1,1,S,S]": torch.where((i[:, None] // 1024) == (i[None, :] // 1024), 0.0, float("-inf")).to(torch.float16)[None, None].contiguous(),
             "all zero [1,1,S,S]": torch.zeros(1, 1, S, S, device=dev, dtype=torch.float16)}
    if B * H * S * S * 2 <= (2 << 30):
        masks["per-head [1,H,S,S]"] = (-(i[:, None] - i[None, :]).abs().float()[None] / (64.0 * (1 + torch.arange(H, device=dev)[:, None, None]))).to(torch.float16)[None].contiguous()
    t0 = graph_ms(lambda: umfa_torch.attention_forward(q, k, v, out=out))
    for name, m in masks.items():
        row = {"shape": [B, H, S, D], "mask": name, "unmasked_ms": round(t0, 4)}
        for side_, opts in (("w64_bias", {}), ("row128", {"no_w64_bias": 1})):
            with umfa_torch.options(**opts):
                row[side_ + "_ms"] = round(graph_ms(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=out)), 4)
                row[side_ + "_kernel"] = umfa_torch.last_kernel()
        print(json.dumps(row), flush=True)
    del masks
