#!/usr/bin/env python3
"""random launch sizes (self- and cross-attention, both head dims, causal or not): the dispatcher's choice against the forced alternatives
(force_w64 where that kernel can run, no_w64); prints the launches where the choice is more than 5 % behind the best"""
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
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
DT = torch.float16 if "fp16" in sys.argv else torch.bfloat16
bad = 0
for it in range(N):
    D = rng.choice([64, 128])
    causal = rng.random() < 0.35
    Sq = rng.choice([128, 256, 384, 512, 768, 1024, 1280, 1536, 2048, 2304, 3072, 4096, 6144, 8192])
    Skv = Sq if causal or rng.random() < 0.6 else rng.choice([77, 256, 512, 1024, 2048, 4096, 8192, 16384])
    bh_max = max(1, int(6e11 / (4.0 * Sq * Skv * D)))  # <= ~0.6 TFLOP per launch
    BH = min(bh_max, rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128, 256]))
    B = rng.choice([b for b in (1, 2, 4, 8) if BH % b == 0])
    H = BH // B
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=DT)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=DT)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=DT)
    o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv} D{D} {'causal' if causal else 'full'}"}
    for name, opts in (("default", {}), ("w64", {"force_w64": 1}), ("r128", {"no_w64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, causal=causal, out=o), n=20)
            row[name + "_kernel"] = umfa_torch.last_kernel()
    best = min(row["w64_us"], row["r128_us"])
    row["default_over_best"] = round(row["default_us"] / best, 3)
    if row["default_over_best"] > 1.05:
        bad += 1
        row["MISS"] = True
    print(json.dumps(row), flush=True)
print(json.dumps({"launches": N, "more_than_5pct_behind": bad}))
