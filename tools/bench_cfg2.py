#!/usr/bin/env python3
"""BASELINE config 2 (B4 H16 S1024 D64 bf16 causal forward) and neighbours, graph-replayed, with and without the causal
half-split of fa_fwd16 (option no_split)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402


def graph_ms(fn, n=200):
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
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
    return a.elapsed_time(b) / n


res = {}
for (B, H, S, D) in [(4, 16, 1024, 64), (4, 16, 1024, 128), (2, 16, 2048, 64), (8, 16, 512, 64), (1, 32, 2048, 128), (2, 8, 1024, 64)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    row = {}
    for name, opts in (("split", {"force_split": 2}), ("nosplit", {})):
        with umfa_torch.options(**opts):
            row[name] = round(graph_ms(lambda: umfa_torch.attention_forward(q, k, v, causal=True, out=o)) * 1e3, 2)
            row[name + "_kernel"] = umfa_torch.last_kernel()
            row[name + "_o"] = o.clone()
    row["equal_to_1e-2"] = bool((row.pop("split_o").float() - row.pop("nosplit_o").float()).abs().max() < 1e-2)
    fl = 2.0 * B * H * S * S * D
    row["tflops_split"] = round(fl / row["split"] / 1e6, 1)
    res[f"B{B}_H{H}_S{S}_D{D}"] = row
print(json.dumps(res))
