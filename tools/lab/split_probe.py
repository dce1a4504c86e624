#!/usr/bin/env python3
"""fa_fwd16's split-KV paths after the fold protocol change (sc1 stores / loads + relaxed ticket, no fences): graph-replayed us with the split
plan the dispatcher picks, without any split (option no_split), and -- causal shapes -- with the heavy half of the q-blocks cut in two
(lab option force_split = 2); outputs compared."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402


def graph_us(fn, n=100):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        side.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b) / n)
    return round(best * 1e3, 2)


CASES = [  # B, H, Sq, Skv, D, causal
    (4, 16, 1024, 1024, 64, True), (8, 16, 512, 512, 64, True), (2, 16, 2048, 2048, 64, True), (1, 32, 2048, 2048, 128, True), (4, 16, 1024, 1024, 128, True),
    (1, 8, 1024, 4096, 128, False), (1, 16, 512, 8192, 128, False), (8, 32, 1, 8192, 128, False), (4, 32, 1, 8192, 128, False), (1, 4, 128, 16384, 64, False),
]
def main():
    for (B, H, Sq, Skv, D, causal) in CASES:
        torch.manual_seed(0)
        q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
        k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
        v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
        o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
        row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv} D{D} {'causal' if causal else 'full'}"}
        outs = {}
        modes = [("default", {"no_w64": 1}), ("no_split", {"no_w64": 1, "no_split": 1})] + ([("causal_half_split", {"no_w64": 1, "force_split": 2})] if causal else [])
        for name, opts in modes:
            with umfa_torch.options(**opts):
                row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, causal=causal, out=o))
                row[name + "_kernel"] = umfa_torch.last_kernel()
                outs[name] = o.clone()
        for name in outs:
            if name != "no_split":
                row[name + "_vs_no_split_rel"] = float((outs[name] - outs["no_split"]).abs().max() / outs["no_split"].abs().max())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
