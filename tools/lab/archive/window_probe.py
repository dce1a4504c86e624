#!/usr/bin/env python3
"""sliding windows (mask-free window entry) with bf16 default options: dispatcher's choice / forced one-workgroup-per-CU kernel / 128-row kernel"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402

for (B, H, S, D, W, causal) in [(1, 24, 4096, 128, 512, False), (1, 24, 4096, 128, 128, False), (2, 16, 8192, 128, 1024, False), (1, 24, 4096, 128, 512, True), (1, 32, 8192, 128, 256, True),
                                (4, 16, 4096, 128, 256, False), (1, 24, 4096, 64, 512, False), (2, 16, 8192, 64, 1024, True), (1, 16, 16384, 128, 2048, True), (1, 8, 4096, 128, 512, False)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    win = (W, 0 if causal else W)
    row = {"shape": f"B{B} H{H} S{S} D{D} window {win}"}
    for name, opts in (("default", {}), ("w64", {"force_w64": 1}), ("r128", {"no_w64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, window=win, causal=causal, out=o))
            row[name + "_kernel"] = umfa_torch.last_kernel()
    print(json.dumps(row), flush=True)
