#!/usr/bin/env python3
"""Profiling driver: alternate the bf16 forward (fp32 O) and the runtime-quantised int8 forward of one shape, N times each
(python tools/run_pair.py [n] [B H S D]) -- rocprofv3 --kernel-trace --stats then gives the kernel-only comparison."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, S, D = (int(x) for x in sys.argv[2:6]) if len(sys.argv) >= 6 else (1, 24, 4096, 128)
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
for _ in range(n):
    umfa_torch.attention_forward(q, k, v, out=out)
    umfa_torch.quantized_attention_forward_stream(q, k, v)
torch.cuda.synchronize()
print("done")
