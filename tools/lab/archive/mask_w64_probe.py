#!/usr/bin/env python3
"""Lab (round 4): bool mask tensors on the one-wave-per-SIMD kernel (fa_fwd16_w64<.,128,mask>: mask pack pre-pass + tile lists) against
the 128-row kernel's tile-flag path (option no_w64_mask), graph-replayed ms per call, pre-passes included."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402


def graph_ms(fn, n=10):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        side.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) / n)
    return sorted(ts)[len(ts) // 2]


SHAPES = [(1, 24, 4096), (1, 20, 4096), (3, 24, 4096), (2, 16, 4096), (4, 16, 4096), (1, 16, 8192)]
if len(sys.argv) > 1 and sys.argv[1] == "few":  # fewer 256-row blocks than CUs
    SHAPES = [(1, 8, 4096), (1, 12, 4096), (1, 4, 8192), (2, 4, 2048), (1, 2, 16384), (1, 8, 2048)]
for (B, H, S) in SHAPES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, 128, device="cuda", dtype=torch.float32)
    i = torch.arange(S, device="cuda")
    masks = {"padding_73pct [B,1,1,S]": (i < int(S * 0.73))[None, None, None, :].expand(B, 1, 1, S).contiguous(),
             "blockdiag_4 [1,1,S,S]": ((i[:, None] // (S // 4)) == (i[None, :] // (S // 4)))[None, None].contiguous(),
             "window_512 [1,1,S,S]": ((i[:, None] - i[None, :]).abs() <= 512)[None, None].contiguous(),
             "random_80pct [1,H,S,S]": torch.rand(1, H, S, S, device="cuda") < 0.8}
    for name, m in masks.items():
        row = {"shape": f"B{B} H{H} S{S}", "mask": name, "visible": round(float(m.float().mean()), 4)}
        outs = {}
        for route, opts in (("w64", {}), ("r128", {"no_w64_mask": 1})):
            with umfa_torch.options(**opts):
                row[route + "_ms"] = round(graph_ms(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o)), 5)
                row[route + "_kernel"] = umfa_torch.last_kernel()
                outs[route] = o.clone()
        row["max_rel_diff"] = float((outs["w64"] - outs["r128"]).abs().max() / outs["r128"].abs().max())
        row["r128_over_w64"] = round(row["r128_ms"] / row["w64_ms"], 3)
        print(json.dumps(row), flush=True)
    row = {"shape": f"B{B} H{H} S{S}", "mask": "none", "ms": round(graph_ms(lambda: umfa_torch.attention_forward(q, k, v, out=o)), 5), "kernel": umfa_torch.last_kernel()}
    print(json.dumps(row), flush=True)
