import sys
sys.path[:0] = ['/root/repo', '/root/repo/universal-metal-flash-attention_amd', '/root/repo/tools']
import torch, umfa_torch
from bench_window_ab import graph_us
for (B, H, S, D) in ((1, 8, 4224, 128), (2, 8, 1152, 128), (1, 16, 2176, 128), (1, 8, 4224, 64), (1, 8, 3000, 128), (4, 8, 1100, 128)):
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    r = {}
    for tag, opts in (("default", {"no_w64": 1}), ("paired", {"no_w64": 1, "cbal": 1}), ("unpaired", {"no_w64": 1, "cbal": 2})):
        with umfa_torch.options(**opts):
            r[tag] = round(graph_us(lambda: umfa_torch.attention_forward(q, k, v, causal=True, out=o)), 1)
    print((B, H, S, D), r, flush=True)
