#!/usr/bin/env python3
"""Decode-shaped launches, the decode form of the 128-row kernel (option decode_ks = 0) against the plain form (2): hipGraph replays, one process, the two sides
alternating.   python tools/bench_decode_ab.py [force_split values ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import torch
import umfa_torch
from bench_window_ab import graph_us
SHAPES = [(1, 32, 1, 8192, 128), (8, 32, 1, 8192, 128), (1, 32, 1, 32768, 128), (32, 32, 1, 2048, 128), (8, 32, 1, 8192, 64), (4, 32, 8, 8192, 128), (16, 8, 1, 4096, 128),
          (1, 8, 1, 131072, 128), (64, 8, 1, 1024, 128), (1, 8, 4, 8192, 128), (2, 8, 32, 4096, 128), (4, 8, 1, 2048, 128), (1, 64, 1, 4096, 64), (2, 16, 16, 16384, 128)]
forced = [int(x) for x in sys.argv[1:]]
for (B, H, Sq, Skv, D) in SHAPES:
    torch.manual_seed(0)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    fn = lambda: umfa_torch.attention_forward(q, k, v, out=o)  # noqa: E731
    r = {"dec": [], "plain": []}
    for rnd in range(3):
        for tag, ks in (("dec", 0), ("plain", 2)) if rnd % 2 == 0 else (("plain", 2), ("dec", 0)):
            with umfa_torch.options(decode_ks=ks):
                r[tag].append(graph_us(fn))
    line = f"B{B} H{H} Sq{Sq} Skv{Skv} D{D}: decode form {min(r['dec']):7.1f} us   plain form {min(r['plain']):7.1f} us   ratio {min(r['plain']) / min(r['dec']):.2f}"
    for fs in forced:
        with umfa_torch.options(decode_ks=0, force_split=fs):
            line += f"   dec k={fs}: {graph_us(fn):.1f}"
    byts = 2 * B * H * Skv * D * 2
    print(line + f"   ({byts / min(r['dec']) / 1e6:.2f} TB/s)", flush=True)
