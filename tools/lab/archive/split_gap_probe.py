#!/usr/bin/env python3
"""128-row kernel with between half a workgroup and one workgroup per CU (the split plan takes cus / items = 1 part there): no split against two parts"""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402
import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402
for (B, H, Sq, Skv, D) in [(8, 1, 2304, 2304, 128), (8, 5, 768, 8192, 128), (8, 8, 512, 8192, 128), (1, 9, 2048, 2048, 128), (1, 12, 2048, 2048, 128), (1, 14, 2048, 2048, 128), (2, 10, 1024, 4096, 128),
                          (1, 10, 2048, 2048, 64), (1, 15, 2048, 4096, 64), (3, 16, 512, 2048, 128)]:
    torch.manual_seed(0)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv} D{D}", "items128": B * H * ((Sq + 127) // 128)}
    for name, opts in (("plan", {"no_w64": 1}), ("two_parts", {"no_w64": 1, "force_split": 2}), ("three_parts", {"no_w64": 1, "force_split": 3}), ("w64", {"force_w64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o), n=20)
    print(json.dumps(row), flush=True)
