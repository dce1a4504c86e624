#!/usr/bin/env python3
"""Grouped-query decode through umfa_torch.scaled_dot_product_attention: query heads of a KV head packed into the rows of one tile (round 6) against the
zero-copy head views (UMFA_GQA_PACK_ROWS=0) and against a torch fp32 reference.   python tools/bench_gqa_decode.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import torch
import umfa_torch
from bench_window_ab import graph_us
for (B, Hq, Hkv, Sq, Skv, D) in [(8, 32, 8, 1, 8192, 128), (1, 32, 8, 1, 8192, 128), (1, 64, 8, 1, 32768, 128), (16, 32, 4, 1, 4096, 128), (4, 32, 8, 4, 8192, 128), (8, 16, 2, 1, 8192, 64), (32, 32, 8, 1, 2048, 128)]:
    torch.manual_seed(0)
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, Hkv, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    res = {}
    outs = {}
    for tag, env in (("packed_rows", "1"), ("head_views", "0")):
        os.environ["UMFA_GQA_PACK_ROWS"] = env
        fn = lambda: umfa_torch.scaled_dot_product_attention(q, k, v, enable_gqa=True)  # noqa: E731
        outs[tag] = fn()
        res[tag] = round(graph_us(fn), 1)
    g = Hq // Hkv
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(g, 1), v.float().repeat_interleave(g, 1))
    err = {t: float((o.float() - ref).abs().max() / ref.abs().max()) for t, o in outs.items()}
    byts = 2 * B * Hkv * Skv * D * 2
    print(f"B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Skv{Skv} D{D}: packed rows {res['packed_rows']} us ({byts / res['packed_rows'] / 1e6:.2f} TB/s of K+V, rel {err['packed_rows']:.1e})   head views {res['head_views']} us (rel {err['head_views']:.1e})   [{umfa_torch.last_kernel()}]", flush=True)
