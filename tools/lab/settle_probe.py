#!/usr/bin/env python3
"""What does a 20-launch timed region see after different untimed preambles?  (bench.py's driver regime: --steps 20
--warmup 5.)  Each case: preamble, torch.cuda.synchronize(), then ONE timed region of 20 FLUX launches (graph replay or
eager), wall clock between two synchronizes -- exactly bench.py's bracket.  Repeated 5 times per case with a 1 s idle
sleep in between (a fresh 'cold' start each time)."""
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
K = 20


def step():
    umfa_torch.attention_forward(q, k, v, out=out)


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(5):
        step()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K):
            step()
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


def eager():
    for _ in range(K):
        step()


res = {}
for name, pre_n, mode in (("graph_after_1_replay", 1, "graph"), ("graph_after_5_replays", 5, "graph"), ("graph_after_25_replays", 25, "graph"),
                          ("graph_after_100_replays", 100, "graph"), ("graph_after_400_replays", 400, "graph"),
                          ("eager_after_5_steps", 0, "eager5"), ("eager_after_2000_steps", 0, "eager2000"), ("graph_after_2000_eager_steps", 0, "graph_e2000")):
    ts = []
    for rep in range(5):
        time.sleep(1.0)
        if mode == "graph":
            for _ in range(pre_n):
                g.replay()
            ts.append(timed(g.replay))
        elif mode == "eager5":
            for _ in range(5):
                step()
            ts.append(timed(eager))
        elif mode == "eager2000":
            for _ in range(2000):
                step()
            ts.append(timed(eager))
        else:
            for _ in range(2000):
                step()
            ts.append(timed(g.replay))
    res[name] = [round(t, 4) for t in ts]
# the same 20-launch region repeated back to back with only the bracket's synchronize in between: does the sync gap reset the state?
time.sleep(1.0)
for _ in range(100):
    g.replay()
res["back_to_back_regions_after_100_replays"] = [round(timed(g.replay), 4) for _ in range(12)]
print(json.dumps(res))
