#!/usr/bin/env python3
"""Backward timing (library hipEvents, kernels only): python tools/bench_bwd.py [B H S D] [dtype]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
from umfa._ffi import _lib, _check_error
from umfa_torch import ops
B, H, S, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 24, 4096, 128)
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[5] if len(sys.argv) > 5 else "bf16"]
causal = len(sys.argv) > 6 and sys.argv[6] == "causal"
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(4))
o32, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True)
dq = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32); dk = torch.empty_like(dq); dv = torch.empty_like(dq)
dvec = torch.empty(B * H * S, device="cuda", dtype=torch.float32)
torch.cuda.synchronize()
bufs = [ops._DevBuf(t) for t in (do, q, k, v, o32, lse, dq, dk, dv, dvec)]
ts = []
for i in range(6):
    _check_error(_lib.mfa_attention_backward(ops.context(), *(b.handle for b in bufs), B, S, S, H, D, D ** -0.5, causal,
                                             ops._PREC[dt], ops._PREC[dt], False, False, False, False))
    ts.append(umfa_torch.gpu_latency() * 1e3)
ts = sorted(ts[1:])
fl = 10.0 * B * H * S * S * D * (0.5 if causal else 1.0)
print(f"B{B} H{H} S{S} D{D} {dt} causal={int(causal)} backward {ts[len(ts)//2]:.3f} ms  {fl/ts[len(ts)//2]/1e9:.1f} TFLOP/s (10*B*H*S^2*D)  [{umfa_torch.last_kernel()}]")
