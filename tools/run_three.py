#!/usr/bin/env python3
"""bf16, int8 and int8+fp8 forwards of ONE shape, a few launches each (profiling target).  python tools/run_three.py B H S [n]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, S = (int(x) for x in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 5
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty(B, H, S, 128, device="cuda", dtype=torch.float32)
for _ in range(n):
    umfa_torch.attention_forward(q, k, v, out=out)
    umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise")
    umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise_fp8pv")
torch.cuda.synchronize()
