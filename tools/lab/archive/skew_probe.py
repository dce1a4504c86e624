#!/usr/bin/env python3
"""FLUX forward (bf16, lazy) with the folding part of every two-way cut item shortened by `w64_skew` tiles: sustained-state ms
per launch (graph of 20 launches, 30 untimed replays, then 10 timed replays interleaved over the skew values) + bitwise
equality of the outputs across skews is NOT expected (different fold points) -- parity vs skew 0 is reported."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402

B, H, S, D = 1, 24, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
skews = [0, 1, 2, 3, 4, 6, 8]
graphs, outs = {}, {}
side = torch.cuda.Stream()
for sk in skews:
    umfa_torch.set_option("w64_skew", sk)
    out = torch.empty_like(q)
    with torch.cuda.stream(side):
        for _ in range(3):
            umfa_torch.attention_forward(q, k, v, out=out)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                umfa_torch.attention_forward(q, k, v, out=out)
    graphs[sk], outs[sk] = g, out
umfa_torch.set_option("w64_skew", 0)
with torch.cuda.stream(side):
    for _ in range(30):
        graphs[0].replay()
    times = {sk: [] for sk in skews}
    for rnd in range(10):
        for sk in (skews if rnd % 2 == 0 else skews[::-1]):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            graphs[sk].replay()
            b.record()
            b.synchronize()
            times[sk].append(a.elapsed_time(b) / 20)
res = {}
for sk in skews:
    t = sorted(times[sk])
    res[sk] = {"ms_median": round(t[len(t) // 2], 5), "ms_min": round(t[0], 5),
               "max_abs_diff_vs_skew0": float((outs[sk].float() - outs[0].float()).abs().max())}
print(json.dumps(res))
