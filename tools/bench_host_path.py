#!/usr/bin/env python3
"""PCIe-inclusive rate of the synchronous host-buffer entry (mfa_attention_forward on numpy arrays):
FLUX shape bf16, pageable host memory, H2D of Q/K/V + kernel + D2H of fp32 O.  Reported in DESIGN.md only."""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import numpy as np
import umfa

B, H, S, D = 1, 24, 4096, 128
rng = np.random.default_rng(0)
q, k, v = (rng.integers(0, 2 ** 16, size=(B, H, S, D), dtype=np.uint16) & 0xBFFF for _ in range(3))  # finite bf16 bits
with umfa.MFAContext() as ctx:
    for _ in range(2):
        umfa.flash_attention_forward(ctx, q, k, v, input_precision="bf16", intermediate_precision="bf16", layout="bhsd")
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        umfa.flash_attention_forward(ctx, q, k, v, input_precision="bf16", intermediate_precision="bf16", layout="bhsd")
        ts.append(time.perf_counter() - t0)
    kern = ctx.gpu_latency
ts.sort()
fl = 4.0 * B * H * S * S * D
print(f"host-path wall {ts[len(ts)//2]*1e3:.2f} ms ({fl/ts[len(ts)//2]/1e12:.1f} TFLOP/s PCIe-inclusive), kernel-only {kern*1e6:.1f} us")
