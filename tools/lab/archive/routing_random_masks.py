#!/usr/bin/env python3
"""random launch sizes with bool mask tensors (head_dim 128, bf16): the dispatcher's choice against the 128-row kernel's tile-flag path (option
no_w64_mask) and -- where it can run -- the forced mask kernel (force_w64); pre-passes included, graph-replayed us"""
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
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(N):
    Sq = rng.choice([256, 512, 1024, 1536, 2048, 3072, 4096, 8192])
    Skv = Sq if rng.random() < 0.7 else rng.choice([256, 512, 1024, 2048, 4096, 8192])
    bh_max = max(1, int(5e11 / (4.0 * Sq * Skv * 128)))
    BH = min(bh_max, rng.choice([2, 4, 8, 12, 16, 24, 32, 48, 64, 128]))
    B = rng.choice([b for b in (1, 2, 4) if BH % b == 0])
    H = BH // B
    kind = rng.choice(["padding", "padding", "blockdiag", "window", "random", "causal_pad"])
    i = torch.arange(Sq, device="cuda")[:, None]
    j = torch.arange(Skv, device="cuda")[None, :]
    if kind == "padding":
        lens = torch.tensor([int(Skv * rng.uniform(0.4, 1.0)) for _ in range(B)], device="cuda")
        m = (j[None] < lens[:, None, None])[:, None]
    elif kind == "blockdiag":
        nd = rng.choice([2, 4, 8])
        m = ((i * nd // Sq) == (j * nd // Skv))[None, None]
    elif kind == "window":
        w = rng.choice([128, 512, 1024])
        m = ((i * Skv // Sq - j).abs() <= w)[None, None]
    elif kind == "random":
        m = torch.rand(1, H, Sq, Skv, device="cuda") < rng.choice([0.3, 0.8])
        m[..., 0] = True
    else:
        m = ((j <= i * Skv // Sq) & (j < int(Skv * 0.8)))[None, None]
    m = m.contiguous()
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(B, H, Sq, 128, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv}", "mask": kind, "visible": round(float(m.float().mean()), 3)}
    for name, opts in (("default", {}), ("r128", {"no_w64_mask": 1}), ("w64", {"force_w64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o), n=10)
            row[name + "_kernel"] = umfa_torch.last_kernel()
    best = min(row["w64_us"], row["r128_us"])
    row["default_over_best"] = round(row["default_us"] / best, 3)
    if row["default_over_best"] > 1.08:
        bad += 1
        row["MISS"] = True
    print(json.dumps(row), flush=True)
    del m, q, k, v, o
print(json.dumps({"launches": N, "more_than_8pct_behind": bad}))
