#!/usr/bin/env python3
"""one-workgroup-per-CU kernel with slightly fewer items than CUs: stream-K over all CUs (default) against one whole item per workgroup
(lab option w64_grid = items) and the 128-row kernel"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402

SHAPES2 = [(1, 18, 2048, 128), (1, 20, 2048, 128), (1, 22, 2048, 128), (1, 24, 2048, 128), (1, 11, 4096, 128), (1, 12, 4096, 128), (1, 29, 4096, 128), (2, 15, 4096, 128), (1, 60, 2048, 128),
           (1, 56, 2048, 128), (1, 27, 4096, 128), (1, 58, 2048, 64)]
for (B, H, S, D) in SHAPES2 if len(sys.argv) > 1 else [(1, 24, 2304, 128), (1, 24, 2304, 64), (1, 26, 2048, 128), (1, 28, 2048, 128), (1, 30, 2048, 128), (1, 20, 2560, 128), (1, 14, 4096, 128), (1, 15, 4096, 128),
                     (1, 13, 4096, 64), (1, 22, 2560, 64)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    items = B * H * ((S + 255) // 256)
    row = {"shape": f"B{B} H{H} S{S} D{D}", "items": items}
    whole_opts = {"force_w64": 1, "w64_grid": items} if items <= 256 else {"force_w64": 1}
    for name, opts in (("streamk", {"force_w64": 1, "w64_skew": 1} if items > 256 else {"force_w64": 1, "w64_grid": 256}), ("whole", whole_opts), ("r128", {"no_w64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o))
            row[name + "_kernel"] = umfa_torch.last_kernel()
    print(json.dumps(row), flush=True)
