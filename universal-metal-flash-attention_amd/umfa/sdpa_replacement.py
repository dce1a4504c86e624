"""`MetalSDPA`: the reference's CPU-tensor drop-in for F.scaled_dot_product_attention
(examples/pytorch_sdpa_replacement.py:48-139), over this build's C ABI.

Call contract kept: a callable object holding one MFAContext; arguments (query, key, value, attn_mask, dropout_p,
is_causal, scale); torch tensors on any device go torch -> numpy -> `umfa.flash_attention_forward`
(`mfa_attention_forward` on host arrays) -> torch on the original device in the original dtype; 2-D [S, D] operands
and 4-D [B, S, 1, D] operands (the umfa layout, single head) as in the reference; fp32 stays fp32, fp16 stays fp16,
anything else is computed as fp16 (:110-117).  `attn_mask` and `dropout_p` are accepted and ignored with a warning,
like the reference (:89-96) -- use is_causal.  This is BASELINE config 1's path (B1 H1 S128 D64 fp32): plumbing,
PCIe-inclusive, never the measured one.  Extension over the reference: layout="bhsd" accepts multi-head
[B, H, S, D] tensors (the reference's wrapper is single-head only, core.py:336-340).
"""
from __future__ import annotations

import warnings
from typing import Optional

import numpy as np

from .core import MFAContext, flash_attention_forward
from .utils import is_device_available


class MetalSDPA:
    def __init__(self, layout: str = "bshd"):
        if not is_device_available():
            raise RuntimeError("MetalSDPA: no gfx950 device (there is no CPU fallback)")
        self.context = MFAContext()
        self.layout = layout

    def close(self) -> None:
        self.context.close()

    def __call__(self, query, key, value, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False,
                 scale: Optional[float] = None):
        import torch
        if dropout_p > 0:
            warnings.warn("MetalSDPA: dropout is not supported and is ignored", stacklevel=2)
        if attn_mask is not None:
            warnings.warn("MetalSDPA: attn_mask is ignored (use is_causal), as in the reference wrapper", stacklevel=2)
        device, dtype = query.device, query.dtype
        arrays = [t.detach().contiguous().cpu() for t in (query, key, value)]
        if dtype == torch.float32:
            precision = "fp32"
        else:
            precision = "fp16"
            arrays = [a.to(torch.float16) for a in arrays]
        q, k, v = (a.numpy() for a in arrays)
        out = flash_attention_forward(self.context, q, k, v, causal=bool(is_causal), softmax_scale=scale,
                                      input_precision=precision, intermediate_precision=precision,
                                      output_precision=precision, layout=self.layout)
        res = torch.from_numpy(np.ascontiguousarray(out))
        if device.type != "cpu":
            res = res.to(device)
        return res if res.dtype == dtype else res.to(dtype)
