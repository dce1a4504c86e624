#!/usr/bin/env python3
"""Additive masks of ragged shapes: the bias kernels through the padded copy against the 128-row kernel (option no_w64_ragged_mask).  Graph-replayed; JSON lines."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import torch
import umfa_torch
from bench_mask_f32 import graph_us

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for (B, H, S, D) in [(1, 24, 4000, 128), (1, 24, 4097, 128), (2, 16, 3001, 64), (4, 16, 2040, 128), (8, 8, 1500, 128), (4, 16, 1111, 128)]:
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    i = torch.arange(S, device="cuda")
    d = (i[:, None] - i[None, :]).abs().float()
    lens = torch.tensor([S - (S // (4 * B)) * b for b in range(B)], device="cuda")
    masks = {"rel-pos bias [1,1,S,S] fp16": (-d / 256.0).to(torch.float16)[None, None].contiguous(),
             "rel-pos bias [1,1,S,S] bf16": (-d / 256.0).to(torch.bfloat16)[None, None].contiguous(),
             "rel-pos bias [1,1,S,S] fp32 (held by fp16)": (-d / 256.0).to(torch.float16).float()[None, None].contiguous(),
             "causal + padding 0 / finfo.min [B,1,S,S] bf16": torch.where((i[None, :, None] >= i[None, None, :]) & (i[None, None, :] < lens[:, None, None]), 0.0,
                                                                          torch.finfo(torch.bfloat16).min).to(torch.bfloat16)[:, None].contiguous()}
    for name, m in masks.items():
        t = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
        kern = umfa_torch.last_kernel().split(" (")[0]
        with umfa_torch.options(no_w64_ragged_mask=1):
            t2 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
            k2 = umfa_torch.last_kernel().split(" (")[0]
        rec = {"shape": f"B{B} H{H} S{S} D{D}", "mask": name, "us": round(t, 1), "kernel": kern, "row128_us": round(t2, 1), "row128_kernel": k2}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
    t0 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o))
    print(json.dumps({"shape": f"B{B} H{H} S{S} D{D}", "mask": "none", "us": round(t0, 1), "kernel": umfa_torch.last_kernel()}), flush=True)
