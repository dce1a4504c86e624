#!/usr/bin/env python3
"""the dS-store form of the backward (option bwd_ds_store) against the recomputing form: gradients (both against fp64 autograd
on a small shape, and against each other at FLUX) and time (graph replays interleaved)"""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch


def grads(q, k, v, do, ds_store, causal=False):
    with umfa_torch.options(bwd_ds_store=1 if ds_store else 0):
        o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True)
        g = umfa_torch.attention_backward(do, q, k, v, o, lse, scale=q.shape[-1] ** -0.5, causal=causal)
        return [t.clone() for t in g], umfa_torch.last_kernel()


for dt in (torch.bfloat16, torch.float16):
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(2, 3, 512, 128, device="cuda", dtype=dt) for _ in range(4))
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    s = torch.matmul(qr, kr.transpose(-1, -2)) * 128 ** -0.5
    torch.matmul(torch.softmax(s, -1), vr).backward(do.double())
    for mode in (False, True):
        g, kern = grads(q, k, v, do, mode)
        rel = [float((a.double() - b.grad).abs().max() / b.grad.abs().max()) for a, b in zip(g, (qr, kr, vr))]
        print(str(dt), "ds_store" if mode else "recompute", kern, "rel dq dk dv", ["%.2e" % r for r in rel], flush=True)

torch.manual_seed(1)
B, H, S, D = 1, 24, 4096, 128
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
ga, _ = grads(q, k, v, do, False)
gb, _ = grads(q, k, v, do, True)
print("FLUX ds_store vs recompute: max abs diff dq dk dv", [float((a.float() - b.float()).abs().max()) for a, b in zip(ga, gb)],
      "max |grad|", [float(a.float().abs().max()) for a in ga], flush=True)
side = torch.cuda.Stream()
graphs = {}
for mode in (0, 1):
    umfa_torch.set_option("bwd_ds_store", mode)
    with torch.cuda.stream(side):
        for _ in range(3):
            umfa_torch.attention_backward(do, q, k, v, o, lse, scale=D ** -0.5)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(10):
                umfa_torch.attention_backward(do, q, k, v, o, lse, scale=D ** -0.5)
    graphs[mode] = g
umfa_torch.set_option("bwd_ds_store", 0)
with torch.cuda.stream(side):
    for _ in range(5):
        for m in graphs:
            graphs[m].replay()
    times = {m: [] for m in graphs}
    for rnd in range(9):
        for m in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            graphs[m].replay()
            b.record()
            b.synchronize()
            times[m].append(a.elapsed_time(b) / 10)
fl = 2.5 * 4 * B * H * S * S * D
res = {("ds_store" if m else "recompute"): {"ms_median": round(sorted(t)[len(t) // 2], 5), "frac": round(fl / sorted(t)[len(t) // 2] / 1e9 / 2500, 4)} for m, t in times.items()}
print(json.dumps(res))
