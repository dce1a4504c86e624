#!/usr/bin/env python3
"""does a capture-private scratch pool go away with its graph?  capture / replay / delete a FLUX-shape forward 40 times, an eager
call after each (the reaping point); free device memory per iteration"""
import gc
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

q, k, v = (torch.randn(1, 24, 4096, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
side = torch.cuda.Stream()
used = []
for it in range(40):
    with torch.cuda.stream(side):
        umfa_torch.attention_forward(q, k, v, out=out)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            umfa_torch.attention_forward(q, k, v, out=out)
    g.replay()
    torch.cuda.synchronize()
    del g
    gc.collect()
    umfa_torch.attention_forward(q, k, v, out=out)  # eager call on the current stream: reaps pools whose graphs are gone
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    used.append((total - free) / 2 ** 20)
print("device MiB in use after iterations 1, 5, 10, 20, 40:", [round(used[i]) for i in (0, 4, 9, 19, 39)])
print("growth 5 -> 40: %.0f MiB" % (used[39] - used[4]))
