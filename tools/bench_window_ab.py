#!/usr/bin/env python3
"""Windows and structured bool masks: the one-wave-per-SIMD kernel against the 128-row kernel, graph-replayed in one process.
    python tools/bench_window_ab.py  ->  JSON lines {case, shape, default_us, default_kernel, w64_us, r128_us}"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch


def graph_us(fn, n=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        g.replay(); side.synchronize()
        best = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); b.synchronize()
            best.append(a.elapsed_time(b) / n * 1e3)
    torch.cuda.current_stream().wait_stream(side)
    return sorted(best)[len(best) // 2]


def main():
    shapes = [(1, 24, 4096, 128), (2, 16, 4096, 64), (1, 16, 8192, 128), (4, 16, 2048, 128)]
    for (B, H, S, D) in shapes:
        torch.manual_seed(0)
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        i = torch.arange(S, device="cuda")
        cases = {"window512": dict(window=(512, 512)), "window256": dict(window=(256, 256)), "window1024": dict(window=(1024, 1024)),
                 "causal_window512": dict(window=(512, 0), causal=True),
                 "blockdiag1024_bool": dict(mask=((i[:, None] // 1024) == (i[None, :] // 1024))[None, None].contiguous()),
                 "window512_bool_tensor": dict(mask=((i[:, None] - i[None, :]).abs() <= 512)[None, None].contiguous())}
        for name, kw in cases.items():
            fn = lambda: umfa_torch.attention_forward(q, k, v, out=o, **kw)  # noqa: E731
            r = {"case": name, "shape": [B, H, S, D]}
            for tag, opts in (("default", {}), ("w64", {"force_w64": 1}), ("r128", {"no_w64": 1})):
                with umfa_torch.options(**opts):
                    r[tag + "_us"] = round(graph_us(fn), 1)
                    r[tag + "_kernel"] = umfa_torch.last_kernel()
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
