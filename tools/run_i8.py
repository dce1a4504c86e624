#!/usr/bin/env python3
"""Profiling driver: runtime-quantised int8 forward, FLUX shape (python tools/run_i8.py [n] [mode])"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "blockwise"
torch.manual_seed(0)
q, k, v = (torch.randn(1, 24, 4096, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for _ in range(n):
    umfa_torch.quantized_attention_forward(q, k, v, quant_mode=mode)
print("gpu latency ms", umfa_torch.gpu_latency() * 1e3, umfa_torch.last_kernel())
