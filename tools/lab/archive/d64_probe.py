#!/usr/bin/env python3
"""head_dim 64: fa_fwd16_w64<.,64> (one wave per SIMD, persistent) against fa_fwd16<.,64> (128-row workgroups), same box, same
process: per shape a graph of 20 launches per kernel, 20 untimed replays, then 9 timed replays interleaved.  One JSON line per
shape: ms per launch (median / min), TFLOP/s, and the max difference between the two kernels' outputs."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402

SHAPES = [(1, 24, 4096, False, "bf16"), (2, 16, 4096, False, "bf16"), (4, 16, 2048, False, "bf16"), (1, 16, 8192, False, "bf16"),
          (8, 16, 1024, False, "bf16"), (1, 48, 4096, False, "bf16"), (4, 32, 4096, False, "bf16"),
          (4, 16, 1024, True, "bf16"), (8, 16, 1024, True, "bf16"), (8, 16, 2048, True, "bf16"), (4, 16, 4096, True, "bf16"),
          (1, 32, 8192, True, "bf16"), (4, 16, 8192, True, "bf16"), (2, 16, 4096, False, "fp16"), (4, 16, 4096, True, "fp16")]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) if x.isdigit() else (x == "True") if x in ("True", "False") else x for x in a.split(",")) for a in sys.argv[1:]]
side = torch.cuda.Stream()
for B, H, S, causal, dn in SHAPES:
    dt = torch.bfloat16 if dn == "bf16" else torch.float16
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 64, device="cuda", dtype=torch.float32).to(dt) for _ in range(3))
    graphs, outs, names = {}, {}, {}
    for which in ("w64", "r128"):
        umfa_torch.set_option("force_w64", 1 if which == "w64" else 0)
        umfa_torch.set_option("no_w64", 0 if which == "w64" else 1)
        out = torch.empty_like(q)
        with torch.cuda.stream(side):
            for _ in range(3):
                umfa_torch.attention_forward(q, k, v, causal=causal, out=out)
            names[which] = umfa_torch.last_kernel()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(20):
                    umfa_torch.attention_forward(q, k, v, causal=causal, out=out)
        graphs[which], outs[which] = g, out
    with torch.cuda.stream(side):
        for _ in range(10):
            graphs["w64"].replay()
            graphs["r128"].replay()
        times = {w: [] for w in graphs}
        for rnd in range(9):
            for w in (("w64", "r128") if rnd % 2 == 0 else ("r128", "w64")):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                graphs[w].replay()
                b.record()
                b.synchronize()
                times[w].append(a.elapsed_time(b) / 20)
    flops = 4.0 * B * H * S * S * 64 * (0.5 if causal else 1.0)
    rec = {"shape": [B, H, S, 64], "causal": causal, "dtype": dn}
    for w in graphs:
        t = sorted(times[w])
        rec[w] = {"kernel": names[w], "ms_median": round(t[len(t) // 2], 5), "ms_min": round(t[0], 5),
                  "tflops": round(flops / t[len(t) // 2] / 1e9, 1)}
    rec["speedup_w64"] = round(rec["r128"]["ms_median"] / rec["w64"]["ms_median"], 3)
    rec["max_abs_diff"] = float((outs["w64"].float() - outs["r128"].float()).abs().max())
    print(json.dumps(rec), flush=True)
umfa_torch.set_option("force_w64", 0)
umfa_torch.set_option("no_w64", 0)
