#!/usr/bin/env python3
"""runtime-quantised forward (int8 block-wise), bf16 inputs: the dispatcher's choice against the 128-row int8 kernel (option no_w64) and, where the one-wave-per-SIMD
kernel runs with fewer items than CUs, against cutting every item (lab option w64_grid = 256); quantiser included, graph-replayed us"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402

for (B, H, S) in [(1, 24, 4096), (1, 16, 8192), (1, 12, 4096), (1, 8, 4096), (8, 16, 1024), (1, 24, 2304), (1, 14, 4096), (2, 8, 2048), (1, 4, 8192), (16, 16, 512)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, 128, device="cuda", dtype=torch.float32)
    lse = torch.empty(B * H * S, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} S{S} D128", "items": B * H * ((S + 255) // 256)}
    for name, opts in (("default", {}), ("r128", {"no_w64": 1}), ("cut256", {"w64_grid": 256})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, out=o, lse=lse), n=20)
            row[name + "_kernel"] = umfa_torch.last_kernel()
    print(json.dumps(row), flush=True)
