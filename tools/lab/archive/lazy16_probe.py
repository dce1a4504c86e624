#!/usr/bin/env python3
"""fp16 P under the lazy softmax reference (fp16 kernels and the int8 kernel): lazy (default) against deferred (tau = 6), same
process, graph replays interleaved; rel-err of both against the oracle on sampled rows."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from oracle import oracle, parity  # noqa: E402

CASES = [("fp16", 1, 24, 4096, 128, False), ("fp16", 1, 16, 8192, 128, False), ("fp16", 4, 16, 4096, 128, True), ("fp16", 2, 16, 4096, 64, False),
         ("int8", 1, 24, 4096, 128, False), ("int8", 1, 16, 8192, 128, False), ("int8", 4, 16, 4096, 128, True)]
side = torch.cuda.Stream()
for kind, B, H, S, D, causal in CASES:
    torch.manual_seed(0)
    dt = torch.float16 if kind == "fp16" else torch.bfloat16
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(dt) for _ in range(3))
    graphs, outs, names = {}, {}, {}
    for mode in ("default", "deferred"):
        umfa_torch.set_option("softmax_reference", mode)
        out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)

        def call():
            if kind == "fp16":
                return umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, out=out)
            return umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, quant_mode="blockwise")
        with torch.cuda.stream(side):
            for _ in range(3):
                r = call()
            names[mode] = umfa_torch.last_kernel()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(10):
                    call()
        graphs[mode], outs[mode] = g, (r if isinstance(r, torch.Tensor) else r[0])
    umfa_torch.set_option("softmax_reference", "default")
    with torch.cuda.stream(side):
        for _ in range(5):
            for m in graphs:
                graphs[m].replay()
        times = {m: [] for m in graphs}
        for rnd in range(9):
            for m in (list(graphs) if rnd % 2 == 0 else list(graphs)[::-1]):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                graphs[m].replay()
                b.record()
                b.synchronize()
                times[m].append(a.elapsed_time(b) / 10)
    rec = {"kind": kind, "shape": [B, H, S, D], "causal": causal}
    rows = parity.sample_rows(S)
    qn, kn, vn = (t.cpu().view(torch.int16).numpy().view(np.uint16) if t.dtype == torch.bfloat16 else t.cpu().numpy() for t in (q, k, v))
    ref = oracle.sdpa_forward_rows(qn, kn, vn, rows, causal=causal) if kind == "fp16" else None
    for m in graphs:
        t = sorted(times[m])
        rec[m] = {"kernel": names[m], "ms_median": round(t[len(t) // 2], 5), "ms_min": round(t[0], 5)}
        if ref is not None:
            o = outs[m][:, :, rows].cpu().numpy().astype(np.float64)
            rec[m]["rel"] = float(np.abs(o - ref).max() / np.abs(ref).max())
            rec[m]["rms"] = float(np.sqrt(((o - ref) ** 2).mean() / (ref.astype(np.float64) ** 2).mean()))
    rec["speedup_lazy"] = round(rec["deferred"]["ms_median"] / rec["default"]["ms_median"], 4)
    rec["max_abs_diff_lazy_vs_deferred"] = float((outs["default"].float() - outs["deferred"].float()).abs().max())
    print(json.dumps(rec), flush=True)
