#!/usr/bin/env python3
"""128-row kernel, bf16 default arithmetic, head_dim 128 non-causal: 32-key tiles (three workgroups per CU) against 64-key tiles (option bn64)"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402

for (B, H, Sq, Skv) in [(16, 16, 512, 512), (8, 32, 256, 8192), (8, 32, 512, 4096), (4, 32, 768, 2048), (1, 8, 2048, 2048), (1, 2, 4096, 4096), (16, 16, 256, 256), (1, 24, 4096, 77),
                        (1, 24, 4096, 4096), (2, 8, 1024, 1024), (8, 32, 1, 8192), (1, 16, 512, 8192)]:
    torch.manual_seed(0)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(B, H, Sq, 128, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv}"}
    outs = {}
    for name, opts in (("bn32", {"no_w64": 1}), ("bn64", {"no_w64": 1, "bn64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o))
            row[name + "_kernel"] = umfa_torch.last_kernel()
            outs[name] = o.clone()
    row["rel_diff"] = float((outs["bn32"] - outs["bn64"]).abs().max() / outs["bn64"].abs().max())
    row["bn64_over_bn32"] = round(row["bn64_us"] / row["bn32_us"], 3)
    print(json.dumps(row), flush=True)
