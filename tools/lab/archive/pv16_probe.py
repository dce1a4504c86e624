#!/usr/bin/env python3
"""Lab (round 4): the default bf16 forward (fp16 P V, V converted in the kernel) against the bf16 P V kernels (pv_fp16 = 0):
graph-replayed ms per launch in the sustained power state, interleaved, and rel-err against the oracle on a row subset."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from oracle import oracle, parity  # noqa: E402


def make_graph(fn, n):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    return g, side


def time_graph(g, side, n, reps=5):
    out = []
    with torch.cuda.stream(side):
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            out.append(a.elapsed_time(b) / n)
    return out


SHAPES = [(1, 24, 4096, 128, False), (1, 16, 8192, 128, False), (4, 16, 4096, 128, True), (2, 16, 4096, 64, False), (4, 16, 4096, 64, True),
          (1, 4, 32768, 128, False)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(",")[:4]) + (a.endswith("c"),) for a in sys.argv[1:]]
for (B, H, S, D, causal) in SHAPES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
    n = 20 if S <= 8192 else 4
    for odt in (torch.float32, torch.bfloat16):
        o = torch.empty(B, H, S, D, device="cuda", dtype=odt)
        gs = {}
        for pv in (1, 0):
            umfa_torch.set_option("pv_fp16", pv)
            gs[pv] = make_graph(lambda: umfa_torch.attention_forward(q, k, v, causal=causal, out=o), n) + (umfa_torch.last_kernel(),)
        for _ in range(15):  # settle: the board's sustained power state
            for pv in (1, 0):
                gs[pv][0].replay()
        torch.cuda.synchronize()
        ts = {1: [], 0: []}
        for _ in range(4):
            for pv in (1, 0):
                ts[pv] += time_graph(gs[pv][0], gs[pv][1], n, reps=3)
        row = {"shape": f"B{B} H{H} S{S} D{D}{' causal' if causal else ''}", "out": str(odt).split(".")[-1]}
        for pv in (1, 0):
            row[f"pv{pv}_ms"] = round(float(np.median(ts[pv])), 5)
            row[f"pv{pv}_min"] = round(float(np.min(ts[pv])), 5)
            row[f"pv{pv}_kernel"] = gs[pv][2]
        row["pv1_over_pv0"] = round(row["pv1_ms"] / row["pv0_ms"], 4)
        if odt == torch.float32 and S <= 8192:
            rows = parity.sample_rows(S)
            ref = oracle.sdpa_forward_rows(parity.bits(q), parity.bits(k), parity.bits(v), rows, causal=causal).astype(np.float64)
            for pv in (1, 0):
                umfa_torch.set_option("pv_fp16", pv)
                oo = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32)
                torch.cuda.synchronize()
                d = oo[:, :, rows].cpu().numpy().astype(np.float64) - ref
                row[f"pv{pv}_rel"] = float(np.abs(d).max() / np.abs(ref).max())
            row["status"] = umfa_torch.pv_fp16_status()
        print(json.dumps(row), flush=True)
        del gs
    umfa_torch.set_option("pv_fp16", 1)
