#!/usr/bin/env python3
"""Lab: the default bf16 forward and the quantised forward on a stream that owns FOUR CUs (hipExtStreamCreateWithCUMask): a slab of the V cast
pass has 64 workgroups here, about 20 fit on four CUs at once -- the slab's exchange cannot complete by co-residency, the bounded wait has to.
Results must equal the ordinary stream's bit for bit.   python tools/lab/cu_mask_probe.py"""
import ctypes
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

hip = ctypes.CDLL("libamdhip64.so")
stream = ctypes.c_void_p()
mask = (ctypes.c_uint32 * 8)(0x0000000F, 0, 0, 0, 0, 0, 0, 0)  # CUs 0 ... 3
rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), 8, mask)
assert rc == 0, rc
ext = torch.cuda.ExternalStream(stream.value)
torch.manual_seed(0)
B, H, S, D = 1, 2, 16384, 128
q = torch.randn(B, H, 1024, D, device="cuda", dtype=torch.bfloat16)
k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
v[0, 1] *= 1e-6
with umfa_torch.options(force_w64=1):
    ref = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    kern = umfa_torch.last_kernel()
    qref = umfa_torch.quantized_attention_forward_stream(q, k, v)
    qkern = umfa_torch.last_kernel()
    torch.cuda.synchronize()
    with torch.cuda.stream(ext):
        for name, fn, r in (("bf16 forward", lambda: umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32), ref),
                            ("quantised forward", lambda: umfa_torch.quantized_attention_forward_stream(q, k, v), qref)):
            t = time.time()
            o = fn()
            ext.synchronize()
            dt = (time.time() - t) * 1e3
            t = time.time()
            o2 = fn()
            ext.synchronize()
            print(f"{name} on a 4-CU stream: first call {dt:.1f} ms, second {(time.time() - t) * 1e3:.1f} ms, equal to the full-chip stream's result: {bool(torch.equal(o, r))} {bool(torch.equal(o2, r))}")
print("kernels:", kern, qkern)
