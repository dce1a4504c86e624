#!/usr/bin/env python3
"""Sliding windows without a mask tensor: fa_fwd16_w64<.,128,window> against fa_fwd16<.,128> (the 128-row kernel's window
path), same process, graph replays interleaved; full attention of the same shape beside them."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402

CASES = [(1, 24, 4096, (512, 512), False), (1, 24, 4096, (256, 256), False), (1, 24, 4096, (1024, 0), True), (1, 16, 8192, (1024, 1024), False),
         (1, 8, 16384, (1024, 1024), False), (1, 32, 32768, (4096, 0), True), (4, 16, 2048, (128, 128), False)]
side = torch.cuda.Stream()
for B, H, S, win, causal in CASES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
    graphs, outs, names = {}, {}, {}
    for which in ("w64", "r128", "full"):
        umfa_torch.set_option("no_w64", 1 if which == "r128" else 0)
        out = torch.empty_like(q)
        kw = dict(causal=causal, out=out) if which == "full" else dict(causal=causal, window=win, out=out)
        with torch.cuda.stream(side):
            for _ in range(3):
                umfa_torch.attention_forward(q, k, v, **kw)
            names[which] = umfa_torch.last_kernel()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(10):
                    umfa_torch.attention_forward(q, k, v, **kw)
        graphs[which], outs[which] = g, out
    umfa_torch.set_option("no_w64", 0)
    with torch.cuda.stream(side):
        for _ in range(5):
            for w in graphs:
                graphs[w].replay()
        times = {w: [] for w in graphs}
        for rnd in range(9):
            for w in (list(graphs) if rnd % 2 == 0 else list(graphs)[::-1]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                graphs[w].replay()
                b.record()
                b.synchronize()
                times[w].append(a.elapsed_time(b) / 10)
    rec = {"shape": [B, H, S, 128], "window": list(win), "causal": causal}
    for w in graphs:
        t = sorted(times[w])
        rec[w] = {"kernel": names[w], "ms_median": round(t[len(t) // 2], 5), "ms_min": round(t[0], 5)}
    rec["speedup_w64_vs_128row"] = round(rec["r128"]["ms_median"] / rec["w64"]["ms_median"], 3)
    rec["max_abs_diff"] = float((outs["w64"].float() - outs["r128"].float()).abs().max())
    print(json.dumps(rec), flush=True)
