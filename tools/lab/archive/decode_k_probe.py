#!/usr/bin/env python3
"""decode-like launches (one 128-row q-block per head): time against the number of key-range parts"""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402
import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402
for (B, H, Sq, Skv, D) in [(1, 32, 1, 8192, 128), (4, 8, 1, 16384, 128), (8, 4, 16, 32768, 128), (4, 16, 8, 4096, 128), (1, 8, 1, 131072, 128), (1, 32, 1, 32768, 128), (1, 4, 1, 8192, 128),
                          (8, 16, 8, 4096, 128), (16, 16, 1, 2048, 64), (2, 32, 64, 2048, 128), (1, 16, 1, 65536, 64), (4, 32, 1, 8192, 128)]:
    torch.manual_seed(0)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv} D{D}", "items": B * H, "ntiles": (Skv + 63) // 64}
    with umfa_torch.options(no_w64=1):
        row["plan"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o), n=10)
    for kk in (2, 3, 4, 6, 8, 10, 12, 16, 24, 32):
        if kk * 4 > row["ntiles"]:
            continue
        with umfa_torch.options(no_w64=1, force_split=kk):
            row[f"k{kk}"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o), n=10)
    print(json.dumps(row), flush=True)
