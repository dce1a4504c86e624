#!/usr/bin/env python3
"""Where do the ~7 us per step between the kernel's own duration (rocprof: 169.6 us) and bench.py's ms_per_step (177) go?
One 20-step region between two synchronizes (bench.py's bracket), sustained state (60 untimed regions first), interleaved:
  graph        one hipGraph of 20 kernel nodes, replayed            (bench.py today)
  eager        20 calls of umfa_torch.attention_forward(out=)
  graph_K100   the same bracket around a 100-node graph, /100        (fixed cost of the bracket amortised 5x better)
and HIP events recorded INSIDE the bracket around the same work (device-side span of the 20 steps)."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402

B, H, S, D = 1, 24, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)


def step():
    umfa_torch.attention_forward(q, k, v, out=out)


def capture(n):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            step()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                step()
    torch.cuda.synchronize()
    return g


g20, g100 = capture(20), capture(100)
for _ in range(5):
    step()


def region(fn, n):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return wall, a.elapsed_time(b) / n


def region_plain(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def eager20():
    for _ in range(20):
        step()


for _ in range(60):
    g20.replay()
torch.cuda.synchronize()
cases = {"graph": (g20.replay, 20), "eager": (eager20, 20), "graph_K100": (g100.replay, 100)}
res = {n: {"wall": [], "events": [], "wall_plain": []} for n in cases}
for rnd in range(12):
    for name, (fn, n) in (list(cases.items()) if rnd % 2 == 0 else list(cases.items())[::-1]):
        w, e = region(fn, n)
        res[name]["wall"].append(w)
        res[name]["events"].append(e)
        res[name]["wall_plain"].append(region_plain(fn, n))
outp = {}
for name, r in res.items():
    outp[name] = {k2: {"median": round(sorted(v2)[len(v2) // 2], 5), "min": round(min(v2), 5)} for k2, v2 in r.items()}
print(json.dumps(outp))
