#!/usr/bin/env python3
"""random decode-like / few-item launches on the 128-row kernel: the split plan's choice against every forced part count 2 ... 8 and no split"""
import json
import random
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402
import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(N):
    D = rng.choice([64, 128, 128])
    Sq = rng.choice([1, 1, 8, 16, 64, 128, 256, 512, 1024])
    Skv = rng.choice([512, 1024, 2048, 4096, 8192, 16384, 32768])
    BH = rng.choice([1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128])
    while BH * ((Sq + 127) // 128) > 300:
        BH //= 2
    B = rng.choice([b for b in (1, 2, 4, 8) if BH % b == 0])
    H = BH // B
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv} D{D}", "items": BH * ((Sq + 127) // 128)}
    with umfa_torch.options(no_w64=1):
        row["plan_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o), n=10)
    with umfa_torch.options(no_w64=1, no_split=1):
        row["k1_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o), n=10)
    best, bk = row["k1_us"], 1
    for kk in (2, 3, 4, 6, 8):
        with umfa_torch.options(no_w64=1, force_split=kk):
            t = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o), n=10)
        row[f"k{kk}_us"] = t
        if t < best:
            best, bk = t, kk
    row["best_k_upto8"] = bk
    row["plan_over_best"] = round(row["plan_us"] / best, 3)
    if row["plan_over_best"] > 1.1:
        bad += 1
        row["MISS"] = True
    print(json.dumps(row), flush=True)
print(json.dumps({"launches": N, "plan_more_than_10pct_behind_best_forced": bad}))
