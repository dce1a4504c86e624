#!/usr/bin/env python3
"""The software-pipelined (fa_fwd16<.,64,pipe>, the default) and key-split (fa_fwd16<.,64,ks2>, option ksplit = 1) forms of the
head_dim-64 forward kernel against the plain four-wave form (no_pipe = 1): same-process graph replays and the distance between the
outputs / from a torch fp32 reference.

    python tools/lab/ksplit_probe.py > gpurun_out/<trip>/ksplit_probe.json
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402


def graph_us(fn, n=200):
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
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b) / n)
    return best * 1e3


def ref(q, k, v, causal):
    qf, kf, vf = q.float(), k.float(), v.float()
    s = qf @ kf.transpose(-1, -2) / q.shape[-1] ** 0.5
    if causal:
        Sq, Sk = s.shape[-2:]
        s = s.masked_fill(torch.ones(Sq, Sk, dtype=torch.bool, device=q.device).triu(1), float("-inf"))
    return torch.softmax(s, -1) @ vf


SHAPES = [  # B, H, Sq, Skv, causal, dtype
    (4, 16, 1024, 1024, True, torch.bfloat16),   # BASELINE config 2
    (4, 16, 1024, 1024, False, torch.bfloat16),
    (4, 16, 1024, 1024, True, torch.float16),
    (8, 16, 512, 512, True, torch.bfloat16),
    (2, 16, 2048, 2048, True, torch.bfloat16),
    (2, 8, 1024, 1024, True, torch.bfloat16),
    (1, 8, 300, 300, True, torch.bfloat16),      # ragged rows and keys
    (1, 8, 77, 1000, False, torch.bfloat16),     # cross attention, ragged
    (1, 4, 1, 333, False, torch.bfloat16),       # decode-like
    (1, 32, 8192, 8192, True, torch.bfloat16),   # long
    (2, 24, 4096, 4096, False, torch.bfloat16),  # long, fills the chip (forced off the one-wave-per-SIMD kernel below)
]
res = {}
for (B, H, Sq, Skv, causal, dt) in SHAPES:
    torch.manual_seed(0)
    q = torch.randn(B, H, Sq, 64, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, 64, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 64, device="cuda", dtype=dt)
    o = torch.empty(B, H, Sq, 64, device="cuda", dtype=torch.float32)
    r = ref(q, k, v, causal) if B * H * Sq * Skv <= (1 << 28) else None
    row = {}
    outs = {}
    for name, opts in (("pipe", {"no_w64": 1}), ("ks2", {"no_w64": 1, "ksplit": 1}), ("four_wave", {"no_w64": 1, "no_pipe": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = round(graph_us(lambda: umfa_torch.attention_forward(q, k, v, causal=causal, out=o)), 2)
            row[name + "_kernel"] = umfa_torch.last_kernel()
            outs[name] = o.clone()
            outs[name + "_lse"] = umfa_torch.attention_forward(q, k, v, causal=causal, return_lse=True)[1].clone()
            if r is not None:
                row[name + "_rel"] = float((o - r).abs().max() / r.abs().max())
    for f in ("pipe", "ks2"):
        row[f + "_vs_four_wave_rel"] = float((outs[f] - outs["four_wave"]).abs().max() / outs["four_wave"].abs().max())
        row[f + "_lse_max_abs_diff"] = float((outs[f + "_lse"] - outs["four_wave_lse"]).abs().max())
    if r is not None:
        s_ = q.float() @ k.float().transpose(-1, -2) / 8.0
        if causal:
            s_ = s_.masked_fill(torch.ones(Sq, Skv, dtype=torch.bool, device="cuda").triu(1), float("-inf"))
        row["pipe_lse_vs_ref"] = float((outs["pipe_lse"] - torch.logsumexp(s_, -1).reshape(outs["pipe_lse"].shape)).abs().max())
        del s_
    row["finite"] = bool(torch.isfinite(outs["ks2"]).all() and torch.isfinite(outs["pipe"]).all())
    row["speedup_ks2"] = round(row["four_wave_us"] / row["ks2_us"], 3)
    row["speedup_pipe"] = round(row["four_wave_us"] / row["pipe_us"], 3)
    res[f"B{B}_H{H}_Sq{Sq}_Skv{Skv}_{'causal' if causal else 'full'}_{str(dt).split('.')[-1]}"] = row
    print(json.dumps({list(res)[-1]: row}), file=sys.stderr, flush=True)
print(json.dumps(res))
