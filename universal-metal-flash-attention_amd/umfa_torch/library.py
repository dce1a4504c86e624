"""Dispatcher-level binding of the SDPA path: `umfa::sdpa_forward` / `umfa::sdpa_backward` custom ops (fake
implementations + autograd registered, so torch.compile(fullgraph=True) keeps them as single graph nodes) and an opt-in
override of `aten::scaled_dot_product_attention` for the CUDA(=ROCm) dispatch keys.

Reference counterpart: TORCH_LIBRARY_IMPL(aten, MPS, m) { m.impl("scaled_dot_product_attention", ...) }
(examples/pytorch-custom-op-ffi/src/metal_sdpa_backend.cpp:3464-3470): there the backend kernel of the MPS key is
replaced; here the same is done for the CUDA key (inference / no-grad) and the AutogradCUDA key (training), which sees
every caller -- F.scaled_dot_product_attention captured before registration, at::scaled_dot_product_attention from C++,
nn.MultiheadAttention's fast path -- not just the Python attribute the monkey-patch replaces.

    import umfa_torch
    umfa_torch.library.override_aten_sdpa(True)        # dispatcher-level, process-wide, undo with (False)
    out = umfa_torch.library.sdpa(q, k, v, is_causal=True)   # the custom op directly

Under torch.compile the routing function (`umfa_torch.scaled_dot_product_attention`, also what `register_backend()`
installs as F.scaled_dot_product_attention) hands over to `torch.ops.umfa.sdpa_forward` as soon as it is being traced:
ctypes calls cannot be traced, an opaque custom op can.  The nine dispatch counters are bumped inside the op's real
implementation, i.e. at run time of the compiled graph, once per executed attention.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops

_SUPPORTED = (torch.float32, torch.float16, torch.bfloat16)


def _bump(name: str, by: int = 1) -> None:
    from . import sdpa as _s
    _s._bump(name, by)


# ---------------------------------------------------------------------------------------------------------------------
# custom ops.  Operands: BHSD tensors [B,H,Sq,D] / [B,H,Skv,D] on the ROCm device, one dtype of (fp32, fp16, bf16), last
# dim contiguous; attn_mask: bool / fp32 / fp16 / bf16 of <= 4 dims broadcastable onto [B,H,Sq,Skv] or None.
@torch.library.custom_op("umfa::sdpa_forward", mutates_args=(), device_types="cuda")
def sdpa_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_mask: Optional[torch.Tensor], is_causal: bool,
                 scale: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """O (in q's dtype, from the kernels' fused cast-back epilogue) and the fp32 log-sum-exp [B*H*Sq]."""
    for t in (q, k, v):
        if t.stride(-1) != 1:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            break
    _bump("total")          # reached only through library.sdpa / a compiled graph: the eager routing counts for itself
    _bump("fp32_instream")  # (a training call is re-classified as fp32_autograd by _setup_context, like the eager routing)
    out, lse = ops.attention_forward(q, k, v, scale=float(scale), causal=bool(is_causal), mask=attn_mask, out_dtype=q.dtype,
                                     return_lse=True)
    return out, lse


@sdpa_forward.register_fake
def _(q, k, v, attn_mask, is_causal, scale):
    B, H, Sq, D = q.shape
    return q.new_empty((B, H, Sq, D)), q.new_empty((B * H * Sq,), dtype=torch.float32)


@torch.library.custom_op("umfa::sdpa_backward", mutates_args=(), device_types="cuda")
def sdpa_backward(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor,
                  lse: torch.Tensor, is_causal: bool, scale: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """dQ, dK, dV in the operand dtype (umfa_attention_backward_stream; no mask: MFABridge.swift:3171-3282 takes none)."""
    return ops.attention_backward(dout.to(q.dtype).contiguous(), q.contiguous(), k.contiguous(), v.contiguous(),
                                  out.contiguous(), lse, scale=float(scale), causal=bool(is_causal))


@sdpa_backward.register_fake
def _(dout, q, k, v, out, lse, is_causal, scale):
    return torch.empty_like(q, memory_format=torch.contiguous_format), torch.empty_like(k, memory_format=torch.contiguous_format), \
        torch.empty_like(v, memory_format=torch.contiguous_format)


def _setup_context(ctx, inputs, output):
    q, k, v, attn_mask, is_causal, scale = inputs
    out, lse = output
    if attn_mask is not None and (q.requires_grad or k.requires_grad or v.requires_grad):
        raise RuntimeError("umfa::sdpa_forward: the dense backward takes no attn_mask (metal_sdpa_backend.cpp:1798-1803); "
                           "the routing function sends masked training calls to torch's native SDPA")
    ctx.save_for_backward(q, k, v, out, lse)
    ctx.is_causal, ctx.scale = bool(is_causal), float(scale)
    if q.requires_grad or k.requires_grad or v.requires_grad:
        # the op body (which cannot see requires_grad) counted this call as fp32_instream: move it, so that the eager and
        # the op routing report the same statistics for the same model
        _bump("fp32_autograd")
        _bump("fp32_instream", -1)


def _backward(ctx, dout, dlse):
    q, k, v, out, lse = ctx.saved_tensors
    dq, dk, dv = torch.ops.umfa.sdpa_backward(dout, q, k, v, out, lse, ctx.is_causal, ctx.scale)
    return dq, dk, dv, None, None, None


sdpa_forward.register_autograd(_backward, setup_context=_setup_context)


def op_supports(q, k, v, attn_mask, dropout_p, needs_grad: bool, is_causal: bool = False) -> bool:
    """Static (trace-time) conditions under which the custom op serves the call; everything else is torch's.
    The same decisions as the eager routing (sdpa.scaled_dot_product_attention), so that a model gives the same numbers and
    the same dispatch statistics compiled and eager: a quantisation mode set through set_quantization_mode is served by
    the eager Function only (the op has no quantised variant), and is_causal together with an attn_mask is torch's error."""
    from . import sdpa as _s
    if _s._quant_precision != _s.QUANT_NONE or (is_causal and attn_mask is not None):
        return False
    if dropout_p > 0.0 or q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
        return False
    if not (q.is_cuda and k.is_cuda and v.is_cuda) or q.dtype not in _SUPPORTED or k.dtype != q.dtype or v.dtype != q.dtype:
        return False
    if q.shape[0] != k.shape[0] or q.shape[1] != k.shape[1] or k.shape != v.shape or q.shape[3] != k.shape[3]:
        return False
    if q.shape[3] == 0 or q.shape[3] > 1024:  # (the reference callers' limit; above 256: fa_fwd_wide / fa_bwd_wide)
        return False
    if attn_mask is not None:
        if needs_grad or attn_mask.dtype not in (torch.bool,) + _SUPPORTED or attn_mask.dim() > 4:
            return False
        try:
            full = (q.shape[0], q.shape[1], q.shape[2], k.shape[2])
            if tuple(torch.broadcast_shapes(tuple(attn_mask.shape), full)) != full:
                return False
        except RuntimeError:
            return False
    return True


def sdpa(query, key, value, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False, scale: Optional[float] = None,
         enable_gqa: bool = False):
    """F.scaled_dot_product_attention's signature over the custom op (traceable: no ctypes, no locks, no .item()).
    Calls the op cannot serve go to torch's own composite implementation."""
    needs_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
    q, k, v = query, key, value
    if q.dim() == 4 and k.dim() == 4 and q.shape[1] != k.shape[1] and k.shape[1] > 0 and q.shape[1] % k.shape[1] == 0:
        g = q.shape[1] // k.shape[1]  # GQA as the reference does it (metal_sdpa_backend.cpp:1694-1702)
        k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
    if not op_supports(q, k, v, attn_mask, dropout_p, needs_grad, bool(is_causal)):
        return native_sdpa(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale,
                           enable_gqa=enable_gqa)
    sm_scale = float(scale) if scale is not None else float(q.shape[-1]) ** -0.5
    return torch.ops.umfa.sdpa_forward(q, k, v, attn_mask, bool(is_causal), sm_scale)[0]


# ---------------------------------------------------------------------------------------------------------------------
# opt-in override of the aten op
_aten_lib = None
_ATEN_OP = torch.ops.aten.scaled_dot_product_attention.default
_COMPOSITE = torch._C.DispatchKeySet(torch._C.DispatchKey.CompositeImplicitAutograd)


def aten_overridden() -> bool:
    return _aten_lib is not None


def native_sdpa(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, enable_gqa=False):
    """torch's own SDPA.  While the aten override is installed the public entry points lead back to us, so the
    composite kernel is invoked directly (redispatch past the backend keys)."""
    if aten_overridden() and not torch.compiler.is_compiling():
        return _ATEN_OP.redispatch(_COMPOSITE, query, key, value, attn_mask, dropout_p, is_causal, scale=scale,
                                   enable_gqa=enable_gqa)
    from . import sdpa as _s
    return _s._native_sdpa(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale,
                           enable_gqa=enable_gqa)


def _aten_kernel(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None, enable_gqa=False):
    from . import sdpa as _s
    return _s.scaled_dot_product_attention(query, key, value, attn_mask, dropout_p, is_causal, scale, enable_gqa)


def override_aten_sdpa(enable: bool = True) -> None:
    """Install / remove our routing as the kernel of aten::scaled_dot_product_attention for the ROCm device:
    key CUDA (grad mode off / no tensor requires grad) and key AutogradCUDA (training: the routing function attaches its
    own autograd node).  CPU tensors and every other backend keep torch's kernels."""
    global _aten_lib
    if enable and _aten_lib is None:
        lib = torch.library.Library("aten", "IMPL")
        lib.impl("scaled_dot_product_attention", _aten_kernel, "CUDA")
        lib.impl("scaled_dot_product_attention", _aten_kernel, "AutogradCUDA")
        _aten_lib = lib
    elif not enable and _aten_lib is not None:
        _aten_lib._destroy()
        _aten_lib = None
