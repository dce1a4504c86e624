#!/usr/bin/env python3
"""mfa_quantized_backward (blocking C-ABI entry, device buffers wrapped zero-copy) at BASELINE config 4's shape: the 16-bit
MFMA engine on fp16 de-quantised operands (default) against the fp32-exact engine (option bwd_exact) -- wall time of the
whole call (quantiser + cast + two backward kernels + the entry's synchronise), median of n."""
import ctypes
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from umfa import _ffi  # noqa: E402

lib = _ffi._lib
B, H, S, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 16, 8192, 128)
n = 7
torch.manual_seed(0)
ctx = umfa_torch.context()
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
o = umfa_torch.quantized_attention_forward_stream(q, k, v)
lse = torch.empty(B * H * S, device="cuda", dtype=torch.float32)
# LSE of the quantised forward: through the blocking entry with lse
dq, dk, dv = (torch.empty(B, H, S, D, device="cuda", dtype=torch.float32) for _ in range(3))


def wrap(t):
    h = _ffi.mfa_buffer_t()
    _ffi._check_error(lib.mfa_buffer_from_mtl_buffer(ctx, ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(), ctypes.byref(h)))
    return h


bufs = {name: wrap(t) for name, t in dict(q=q, k=k, v=v, o=o, do=do, lse=lse, dq=dq, dk=dk, dv=dv).items()}
rc = lib.mfa_quantized_forward_with_lse(ctx, bufs["q"], bufs["k"], bufs["v"], bufs["o"], bufs["lse"], None, B, S, S, H, D,
                                        float(D) ** -0.5, False, 3, 2, 1)
assert rc == 0, rc
res = {"shape": [B, H, S, D], "flops_algorithmic": 2.5 * 4.0 * B * H * S * S * D}
for name, opts in (("mfma_fp16", {}), ("fp32_exact", {"bwd_exact": 1})):
    with umfa_torch.options(**opts):
        ts = []
        for i in range(n if name == "mfma_fp16" else 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = lib.mfa_quantized_backward(ctx, bufs["q"], bufs["k"], bufs["v"], bufs["o"], bufs["do"], bufs["lse"], bufs["dq"], bufs["dk"],
                                            bufs["dv"], None, B, S, S, H, D, float(D) ** -0.5, False, 3, 2, 1)
            ts.append(time.perf_counter() - t0)
            assert rc == 0, rc
        ts.sort()
        res[name] = {"ms": round(ts[len(ts) // 2] * 1e3, 3), "kernel": umfa_torch.last_kernel(), "tflops": round(res["flops_algorithmic"] / ts[len(ts) // 2] / 1e12, 1),
                     "dq_absmax": float(dq.abs().max()), "finite": bool(torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all())}
        res[name + "_dq"] = dq.clone()
a, b = res.pop("mfma_fp16_dq"), res.pop("fp32_exact_dq")
res["rel_mfma_vs_exact_dq"] = float((a - b).abs().max() / b.abs().max())
res["speedup"] = round(res["fp32_exact"]["ms"] / res["mfma_fp16"]["ms"], 2)
print(json.dumps(res))
