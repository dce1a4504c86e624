#!/usr/bin/env python3
"""bf16 forward vs runtime-quantised forward (int8 block-wise = the reference's arithmetic; fp8 P V fast mode), quantiser
included, same HIP events, interleaved rounds in one process.  python tools/bench_quant_modes.py [rounds]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, (B, H, S) in {"flux B1 H24 S4096": (1, 24, 4096), "cfg4 B1 H16 S8192": (1, 16, 8192)}.items():
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out = torch.empty(B, H, S, 128, device="cuda", dtype=torch.float32)
    fns = {"bf16": lambda: umfa_torch.attention_forward(q, k, v, out=out),
           "int8": lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise"),
           "int8+fp8pv": lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise_fp8pv")}
    for f in fns.values():
        for _ in range(3):
            f()
    times = {n: [] for n in fns}
    for r in range(rounds):
        for n, f in (list(fns.items()) if r % 2 == 0 else list(fns.items())[::-1]):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                f()
            b.record()
            torch.cuda.synchronize()
            times[n].append(a.elapsed_time(b) / 10)
    med = {n: sorted(t)[len(t) // 2] for n, t in times.items()}
    print(name, {n: round(t, 4) for n, t in med.items()}, "speedup int8 %.3f fp8pv %.3f" % (med["bf16"] / med["int8"], med["bf16"] / med["int8+fp8pv"]))
