#!/usr/bin/env python3
"""Split-KV plan probe: time a launch with every forced part count (option force_split; 0 = the plan's own choice, no_split = 1 part), graph-replayed,
one process.  python tools/split_plan_probe.py  ->  JSON lines {shape, us: {k: t}, plan_us}"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
from bench_window_ab import graph_us

SHAPES = [(1, 32, 1, 8192, 128), (16, 8, 1, 4096, 128), (1, 8, 1, 131072, 128), (1, 32, 1, 32768, 128), (2, 16, 16, 4096, 64), (8, 32, 1, 8192, 128),
          (1, 2, 4096, 4096, 128), (1, 8, 1024, 1024, 128), (2, 8, 512, 2048, 128), (1, 4, 2048, 2048, 64), (1, 16, 256, 8192, 128), (4, 16, 128, 1024, 64),
          (1, 24, 128, 4096, 128), (1, 1, 8192, 8192, 128), (3, 5, 640, 3000, 128), (1, 40, 1, 16384, 128)]
for (B, H, Sq, Skv, D) in SHAPES:
    torch.manual_seed(0)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    o = torch.empty_like(q)
    fn = lambda: umfa_torch.attention_forward(q, k, v, out=o)  # noqa: E731
    r = {"shape": [B, H, Sq, Skv, D], "us": {}}
    with umfa_torch.options(no_w64=1):
        r["plan_us"] = round(graph_us(fn), 1)
        with umfa_torch.options(no_split=1):
            r["us"]["1"] = round(graph_us(fn), 1)
        for kk in (2, 3, 4, 6, 8, 12, 16, 24, 32):
            if kk > max(1, ((Skv + 63) // 64) // 4):
                continue
            with umfa_torch.options(force_split=kk):
                r["us"][str(kk)] = round(graph_us(fn), 1)
    print(json.dumps(r), flush=True)
