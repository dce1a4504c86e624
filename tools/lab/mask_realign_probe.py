#!/usr/bin/env python3
"""Masks with unaligned rows on the 128-row kernel: the realigned copy against the in-place read (option no_mask_realign).  Graph-replayed; JSON lines."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import torch
import umfa_torch
from bench_mask_f32 import graph_us

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for (B, H, S, D) in [(2, 8, 777, 128), (4, 16, 1001, 64), (8, 8, 513, 128), (1, 8, 2049, 128), (4, 16, 1111, 128)]:
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    i = torch.arange(S, device="cuda")
    d = (i[:, None] - i[None, :]).abs().float()
    lens = torch.tensor([S - (S // (4 * B)) * b for b in range(B)], device="cuda")
    keep = (i[None, :, None] >= i[None, None, :]) & (i[None, None, :] < lens[:, None, None])
    masks = {"rel-pos bias [1,1,S,S] fp16": (-d / 256.0).to(torch.float16)[None, None].contiguous(),
             "rel-pos bias [1,1,S,S] fp32": (-d / 256.0)[None, None].contiguous(),
             "causal + padding bool [B,1,S,S]": keep[:, None].contiguous(),
             "causal + padding 0 / finfo.min [B,1,S,S] bf16": torch.where(keep, 0.0, torch.finfo(torch.bfloat16).min).to(torch.bfloat16)[:, None].contiguous()}
    for name, m in masks.items():
        with umfa_torch.options(no_w64_ragged_mask=1, no_w64_f32_mask=1, no_w64_mask=1):  # (the 128-row kernel in every case)
            t = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
            kern = umfa_torch.last_kernel()
            with umfa_torch.options(no_mask_realign=1):
                t2 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
        rec = {"shape": f"B{B} H{H} S{S} D{D}", "mask": name, "realigned_copy_us": round(t, 1), "in_place_us": round(t2, 1), "kernel": kern}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
