"""scaled_dot_product_attention routing for torch-ROCm tensors.

Counterpart of MetalSDPABackend::scaled_dot_product_attention and its three autograd Functions
(examples/pytorch-custom-op-ffi/src/metal_sdpa_backend.cpp:1643-1904, 2672-2870, 3139-3397) with the
reference's semantics:
  (i)   default scale 1/sqrt(D) when `scale` is None                               (:1827-1835)
  (ii)  2-D / 3-D inputs are promoted to 4-D BHSD and squeezed back                (:1667-1683)
  (iii) GQA (Hq % Hkv == 0) via repeat_interleave of K, V                           (:1694-1702)
  (iv)  non-accelerator tensors, non-4-D, dtype not in {f32,f16,bf16}, mixed dtypes or dropout > 0
        go to torch's native SDPA (counter pytorch_fallback)                        (:1720-1765)
  (v)   the output has the input dtype                                               (:1442-1444)
  (vi)  an all-true bool mask is the same as no mask                                 (:1771-1784)
  (vii) nine dispatch counters                                                       (metal_sdpa_backend.h:666-681)
The kernels read strided masks in-tile, so (vi) needs no host synchronisation here: the all-true check
(`mask.all().item()`, a device sync in the reference) only runs when UMFA_STRIP_TRUE_MASKS=1.
"""
from __future__ import annotations

import ctypes
import os
import threading
from contextlib import contextmanager
from typing import Optional

import torch
import torch.nn.functional as F

from umfa._ffi import MFA_PRECISION_INT4, MFA_PRECISION_INT8, _check_error, _lib

from . import ops

QUANT_NONE, QUANT_INT8, QUANT_INT4 = 0, MFA_PRECISION_INT8, MFA_PRECISION_INT4   # metal_sdpa_backend.h:650-664
QUANT_TENSOR_WISE, QUANT_BLOCK_WISE = 0, 2

_COUNTER_NAMES = ("total", "quantized_autograd", "fp32_autograd", "fp32_direct", "fp32_instream", "rope_instream",
                  "rope_autograd", "pytorch_fallback", "mask_all_true_skipped")
_stats = {k: 0 for k in _COUNTER_NAMES}
_stats_lock = threading.Lock()
_quant_precision = QUANT_NONE
_quant_mode = QUANT_TENSOR_WISE
_native_sdpa = F.scaled_dot_product_attention
_SUPPORTED = (torch.float32, torch.float16, torch.bfloat16)


def _bump(name: str, by: int = 1) -> None:
    with _stats_lock:
        _stats[name] += by


def get_dispatch_stats() -> dict:
    with _stats_lock:
        return dict(_stats)


def reset_dispatch_stats() -> None:
    with _stats_lock:
        for k in _stats:
            _stats[k] = 0


def set_quantization_mode(precision: int = QUANT_NONE, mode: int = QUANT_TENSOR_WISE) -> None:
    """Process-global switch (metal_sdpa_backend.cpp:3420-3427): 0 = off, 3 = INT8, 4 = INT4;
    mode 0 = tensor-wise, 2 = block-wise."""
    global _quant_precision, _quant_mode
    if precision not in (QUANT_NONE, QUANT_INT8, QUANT_INT4):
        raise ValueError("precision must be 0 (off), 3 (INT8) or 4 (INT4)")
    if mode not in (QUANT_TENSOR_WISE, QUANT_BLOCK_WISE):
        raise ValueError("mode must be 0 (tensor-wise) or 2 (block-wise)")
    _quant_precision, _quant_mode = precision, mode


def get_quantization_mode():
    return _quant_precision, _quant_mode


def _sync_for_blocking_abi(t: torch.Tensor) -> None:
    # the synchronous C-ABI entries run on the legacy default stream and block; make torch's current
    # stream visible to them first (the reference calls torch::mps::synchronize(), :2693-2695)
    torch.cuda.current_stream(t.device).synchronize()


def _dev_bufs(*tensors):
    return [ops._DevBuf(t) for t in tensors]


class _FlashAttentionFn(torch.autograd.Function):
    """MetalFlashAttentionFn (:2672-2870): fp32 O + LSE saved, backward through mfa_attention_backward."""

    @staticmethod
    def forward(ctx, q, k, v, causal: bool, scale: float):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        # O in the operand type from the kernel's epilogue; the SAME tensor serves D = rowsum(dO o O) in the backward
        # (the reference keeps a separate fp32 O for that, :2672-2870: twice the activation, one more cast)
        out, lse = ops.attention_forward(q, k, v, scale=scale, causal=causal, out_dtype=q.dtype, return_lse=True)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.causal, ctx.scale = causal, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, o32, lse = ctx.saved_tensors
        # in-stream (no host synchronisation), gradients in the operand type straight from the kernels' epilogues
        dq, dk, dv = ops.attention_backward(dout.to(q.dtype).contiguous(), q, k, v, o32, lse, scale=float(ctx.scale),
                                            causal=bool(ctx.causal))
        return dq, dk, dv, None, None


class _GqaFlashAttentionFn(torch.autograd.Function):
    """Training with grouped K / V heads WITHOUT the reference's repeat_interleave copies (metal_sdpa_backend.cpp:1694-1702):
    the forward reads K / V through head-stride-0 views (the g query heads of a KV head become the heads of a
    (batch x kv-head) slab), the backward reads them in place (umfa_attention_backward_gqa_stream) and sums dK / dV over
    each group inside the library.  Saves the expanded K and V (2 x Hq / Hkv x their size) in the autograd graph."""

    @staticmethod
    def forward(ctx, q, k, v, causal: bool, scale: float):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, Hq, Sq, D = q.shape
        Hkv, Skv = k.shape[1], k.shape[2]
        g = Hq // Hkv
        qv = q.view(B * Hkv, g, Sq, D)
        kv = k.view(B * Hkv, 1, Skv, D).expand(B * Hkv, g, Skv, D)
        vv = v.view(B * Hkv, 1, Skv, D).expand(B * Hkv, g, Skv, D)
        out, lse = ops.attention_forward(qv, kv, vv, scale=scale, causal=causal, out_dtype=q.dtype, return_lse=True)
        out = out.view(B, Hq, Sq, D)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.causal, ctx.scale = causal, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        res = ops.attention_backward_gqa(dout.to(q.dtype).contiguous(), q, k, v, out, lse, scale=float(ctx.scale), causal=bool(ctx.causal))
        if res is None:  # shapes the MFMA backward does not serve: the reference's route, here only in the backward
            g = q.shape[1] // k.shape[1]
            dq, dke, dve = ops.attention_backward(dout.to(q.dtype).contiguous(), q, k.repeat_interleave(g, 1).contiguous(),
                                                  v.repeat_interleave(g, 1).contiguous(), out, lse, scale=float(ctx.scale),
                                                  causal=bool(ctx.causal))
            B, Hkv, Skv, D = k.shape
            res = dq, dke.view(B, Hkv, g, Skv, D).sum(2).to(k.dtype), dve.view(B, Hkv, g, Skv, D).sum(2).to(v.dtype)
        return res[0], res[1], res[2], None, None


class _QuantizedFlashAttentionFn(torch.autograd.Function):
    """MetalQuantizedFlashAttentionFn (:3139-3397): runtime quantisation of Q, K, V; STE backward."""

    @staticmethod
    def forward(ctx, q, k, v, causal: bool, scale: float, precision: int, mode: int, mask):
        if q.dtype not in (torch.float16, torch.bfloat16):
            q, k, v = q.float(), k.float(), v.float()  # (:3157-3171)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        # in-stream (the reference's forward blocks on the GPU, :3142-3267; same numbers)
        o32, lse = ops.quantized_attention_forward_stream(q, k, v, scale=scale, causal=causal, mask=mask,
                                                          bits=4 if precision == QUANT_INT4 else 8,
                                                          quant_mode="blockwise" if mode == QUANT_BLOCK_WISE else "tensor",
                                                          return_lse=True)
        ctx.save_for_backward(q, k, v, o32, lse)
        ctx.mask = mask
        ctx.args = (causal, scale, precision, mode)
        return o32

    @staticmethod
    def backward(ctx, dout):
        q, k, v, o32, lse = ctx.saved_tensors
        causal, scale, precision, mode = ctx.args
        B, H, Sq, D = q.shape
        Skv = k.shape[2]
        dout = dout.to(q.dtype).contiguous()
        dq = torch.empty((B, H, Sq, D), dtype=torch.float32, device=q.device)
        dk = torch.empty((B, H, Skv, D), dtype=torch.float32, device=q.device)
        dv = torch.empty_like(dk)
        m32 = None
        if ctx.mask is not None:
            m32 = torch.zeros((B, H, Sq, Skv), dtype=torch.float32, device=q.device)
            m32 = (m32.masked_fill(~ctx.mask, float("-inf")) if ctx.mask.dtype == torch.bool else m32 + ctx.mask.float()).contiguous()
        _sync_for_blocking_abi(q)
        bufs = _dev_bufs(q, k, v, o32, dout, lse, dq, dk, dv, m32)
        try:
            _check_error(_lib.mfa_quantized_backward(ops.context(), *(b.handle for b in bufs), B, Sq, Skv, H, D,
                                                     float(scale), bool(causal), int(precision), int(mode),
                                                     ops._PREC[q.dtype]))
        finally:
            for b in bufs:
                b.close()
        return dq.to(q.dtype), dk.to(q.dtype), dv.to(q.dtype), None, None, None, None, None


def _gqa_zero_copy(q, k, v, attn_mask, dropout_p, is_causal, scale):
    """Inference GQA without materialising repeat_interleave(K), repeat_interleave(V) (the reference's route,
    metal_sdpa_backend.cpp:1694-1702): the g query heads of a KV head become the heads of a (batch * kv-head) slab whose
    K / V head stride is 0 -- plain strides on the in-stream entry, K / V are read once per group.  Returns None when
    the views do not apply (gradients needed, masks with per-head / per-batch extent, non-collapsible strides)."""
    if (q.dim() != 4 or k.dim() != 4 or v.dim() != 4 or not (q.is_cuda and k.is_cuda and v.is_cuda) or dropout_p > 0.0
            or q.requires_grad or k.requires_grad or v.requires_grad or _quant_precision != QUANT_NONE
            or q.dtype not in _SUPPORTED or k.dtype != q.dtype or v.dtype != q.dtype or k.shape != v.shape
            or q.size(0) != k.size(0) or q.size(3) != k.size(3) or q.size(3) == 0 or q.size(3) > 1024):
        return None
    B, Hq, Sq, D = q.shape
    Hkv, Skv = k.size(1), k.size(2)
    g = Hq // Hkv
    if attn_mask is not None:
        m = attn_mask  # right-aligned onto [B, H, Sq, Skv]: only masks shared by every batch element and head
        if m.dim() > 4 or (m.dim() == 4 and (m.size(0) != 1 or m.size(1) != 1)) or (m.dim() == 3 and m.size(0) != 1):
            return None  # a per-head or per-batch mask would have to be re-viewed too
    for t, h in ((q, Hq), (k, Hkv), (v, Hkv)):
        if t.stride(3) != 1 or (B > 1 and t.stride(0) != h * t.stride(1)):
            return None
    sm_scale = float(scale) if scale is not None else float(D) ** -0.5
    if (attn_mask is None and not is_causal and g * Sq <= 128 and (Sq == 1 or q.stride(1) == Sq * q.stride(2)) and q.stride(1) % 8 == 0
            and os.environ.get("UMFA_GQA_PACK_ROWS", "1") != "0"):
        # decode-like calls (round 6): the g query heads of a KV head become ROWS of one 128-row tile -- [B * Hkv, 1, g * Sq, D] is a plain view of a
        # head-major q -- so a KV head's keys and values are staged once, by one workgroup (and its split-KV parts), instead of once per query head.
        # No mask, not causal: rows of different query heads see the same keys, so nothing else changes.  B8 Hq32 Hkv8 Sq1 Skv8192: see
        # profiles/r6/gqa_decode_pack_rows.txt
        row_stride = q.stride(1) if Sq == 1 else q.stride(2)  # (one row per head: the heads ARE the rows, whatever the size-1 dimension's stride says)
        qp = q.as_strided((B * Hkv, 1, g * Sq, D), (g * q.stride(1), g * q.stride(1), row_stride, 1), q.storage_offset())
        kp = k.as_strided((B * Hkv, 1, Skv, D), (k.stride(1), k.stride(1), k.stride(2), 1), k.storage_offset())
        vp = v.as_strided((B * Hkv, 1, Skv, D), (v.stride(1), v.stride(1), v.stride(2), 1), v.storage_offset())
        _bump("fp32_instream")
        out = ops.attention_forward(qp, kp, vp, scale=sm_scale, causal=False, out_dtype=q.dtype)
        return out.view(B, Hq, Sq, D)
    qv = q.as_strided((B * Hkv, g, Sq, D), (g * q.stride(1), q.stride(1), q.stride(2), 1), q.storage_offset())
    kv = k.as_strided((B * Hkv, g, Skv, D), (k.stride(1), 0, k.stride(2), 1), k.storage_offset())
    vv = v.as_strided((B * Hkv, g, Skv, D), (v.stride(1), 0, v.stride(2), 1), v.storage_offset())
    _bump("fp32_instream")
    out = ops.attention_forward(qv, kv, vv, scale=sm_scale, causal=bool(is_causal), mask=attn_mask, out_dtype=q.dtype)
    return out.view(B, Hq, Sq, D)


def scaled_dot_product_attention(query, key, value, attn_mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0,
                                 is_causal: bool = False, scale: Optional[float] = None, enable_gqa: bool = False):
    if torch.compiler.is_compiling():
        # being traced (torch.compile): ctypes launches, locks and counters cannot be traced -- hand over to the opaque
        # custom op umfa::sdpa_forward (library.py), which runs the same kernels at run time of the compiled graph
        from . import library
        return library.sdpa(query, key, value, attn_mask, dropout_p, is_causal, scale, enable_gqa)
    if key.dim() != value.dim() or key.dim() < 2 or key.size(-2) != value.size(-2):
        raise RuntimeError("UMFA SDPA: key and value must have matching sequence lengths")
    # (ii) promote 2-D / 3-D
    if 2 <= query.dim() < 4 and key.dim() == query.dim():
        q4, k4, v4 = query, key, value
        while q4.dim() < 4:
            q4, k4, v4 = q4.unsqueeze(0), k4.unsqueeze(0), v4.unsqueeze(0)
        out = scaled_dot_product_attention(q4, k4, v4, attn_mask, dropout_p, is_causal, scale, enable_gqa)
        for _ in range(4 - query.dim()):
            out = out.squeeze(0)
        return out
    _bump("total")
    q, k, v = query, key, value

    def fallback():
        _bump("pytorch_fallback")
        m = attn_mask
        if m is not None and m.dtype in (torch.float16, torch.bfloat16):
            m = m.float()
        from . import library  # torch's own SDPA; past our aten override when that is installed
        return library.native_sdpa(query, key, value, attn_mask=m, dropout_p=dropout_p, is_causal=is_causal, scale=scale,
                                   enable_gqa=enable_gqa)

    # masks the kernels cannot read (dtype, rank) or that do not broadcast onto [B, H, Sq, Skv] go to torch, which
    # raises its own error for a mis-shaped one -- the kernels trust shape and strides, so nothing unchecked reaches them
    if attn_mask is not None and q.dim() == 4:
        if attn_mask.dtype not in (torch.bool, torch.float32, torch.float16, torch.bfloat16) or attn_mask.dim() > 4:
            return fallback()
        try:
            if torch.broadcast_shapes(tuple(attn_mask.shape), (q.size(0), q.size(1), q.size(2), k.size(2))) != \
                    (q.size(0), q.size(1), q.size(2), k.size(2)):
                return fallback()
        except RuntimeError:
            return fallback()
    # (iii) GQA
    if q.dim() >= 3 and k.dim() >= 3 and q.size(-3) != k.size(-3):
        hq, hkv = q.size(-3), k.size(-3)
        if hq > hkv and hq % hkv == 0:
            zc = _gqa_zero_copy(q, k, v, attn_mask, dropout_p, is_causal, scale)
            if zc is not None:
                return zc
            if (q.dim() == 4 and k.dim() == 4 and v.dim() == 4 and attn_mask is None and dropout_p == 0.0 and _quant_precision == QUANT_NONE
                    and q.is_cuda and k.is_cuda and v.is_cuda and q.dtype in (torch.float16, torch.bfloat16) and k.dtype == q.dtype
                    and v.dtype == q.dtype and k.shape == v.shape and q.size(0) == k.size(0) and q.size(3) == k.size(3)
                    and q.size(3) in (64, 128, 256) and (q.requires_grad or k.requires_grad or v.requires_grad)):
                # training: no expanded K / V anywhere (forward through stride-0 views, backward in place)
                _bump("fp32_autograd")
                sm = float(scale) if scale is not None else float(q.size(-1)) ** -0.5
                return _GqaFlashAttentionFn.apply(q, k, v, bool(is_causal), sm)
            k = k.repeat_interleave(hq // hkv, -3).contiguous()
            v = v.repeat_interleave(hq // hkv, -3).contiguous()

    unsupported = (not (q.is_cuda and k.is_cuda and v.is_cuda) or q.dim() != 4 or k.dim() != 4 or v.dim() != 4
                   or q.size(0) != k.size(0) or q.size(0) != v.size(0) or q.size(1) != k.size(1)
                   or q.size(1) != v.size(1) or q.size(3) != k.size(3) or q.size(3) != v.size(3)
                   or q.dtype not in _SUPPORTED or k.dtype != q.dtype or v.dtype != q.dtype
                   or q.size(3) > 1024 or q.size(3) == 0)  # (the reference's own limit: metal_sdpa_backend.cpp:1082-1084)
    if unsupported or dropout_p > 0.0:
        return fallback()

    # (vi) masks
    mask = None
    if attn_mask is not None:
        mask = attn_mask
        # An all-true bool mask is a no-op: strip it and take the unmasked fast path (metal_sdpa_backend.cpp:1771-1784: models such as
        # Z-Image always pass an encoder mask).  The test is the reference's own -- mask.all().item(), a host synchronisation --
        # and on by default like there; UMFA_STRIP_TRUE_MASKS=0 keeps the call sync-free (the mask is then read in-tile and its
        # fully open tiles run without mask reads anyway: same numbers, the 128-row kernel instead of the 256-row one).  Never
        # while the stream is being captured (a synchronisation would invalidate the capture).
        if (mask.dtype == torch.bool and mask.numel() > 0 and os.environ.get("UMFA_STRIP_TRUE_MASKS", "1") != "0"
                and not (mask.is_cuda and torch.cuda.is_current_stream_capturing())):
            if bool(mask.all().item()):
                _bump("mask_all_true_skipped")
                mask = None
        if mask is not None and mask.dtype not in (torch.bool, torch.float32, torch.float16, torch.bfloat16):
            return fallback()
        if mask is not None and mask.dim() > 4:
            return fallback()
    # (i) scale
    sm_scale = float(scale) if scale is not None else float(q.size(-1)) ** -0.5

    if _quant_precision != QUANT_NONE:
        _bump("quantized_autograd")
        out = _QuantizedFlashAttentionFn.apply(q, k, v, bool(is_causal), sm_scale, _quant_precision, _quant_mode, mask)
        return out.to(query.dtype)
    if q.requires_grad or k.requires_grad or v.requires_grad:
        if mask is not None:
            return fallback()  # dense backward takes no mask (:1798-1803)
        _bump("fp32_autograd")
        return _FlashAttentionFn.apply(q, k, v, bool(is_causal), sm_scale)
    # inference: in-stream, zero-copy, output directly in the input dtype (v)
    for t in (q, k, v):
        if t.stride(-1) != 1:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            break
    _bump("fp32_instream")
    return ops.attention_forward(q, k, v, scale=sm_scale, causal=bool(is_causal), mask=mask, out_dtype=q.dtype)


_registered = False


def register_backend() -> None:
    """Route torch.nn.functional.scaled_dot_product_attention through this module (the reference overrides the
    aten op for the MPS key, :3464-3470; its examples/flux harness monkey-patches F.sdpa the same way)."""
    global _registered
    if not _registered:
        F.scaled_dot_product_attention = scaled_dot_product_attention
        _registered = True


def unregister_backend() -> None:
    global _registered
    if _registered:
        F.scaled_dot_product_attention = _native_sdpa
        _registered = False


@contextmanager
def use_umfa_sdpa():
    """Context manager form (reference: pytorch_custom_op_ffi.backend.use_metal_sdpa)."""
    was = _registered
    register_backend()
    try:
        yield
    finally:
        if not was:
            unregister_backend()


# ------------------------------------------------------------------------------------------------------------------
# RoPE + SDPA (MetalSDPABackend::rope_scaled_dot_product_attention, metal_sdpa_backend.cpp:1472-1641; autograd
# MetalRopeFlashAttentionFn :2883-3133): rotate Q and K in-stream with the rotary kernel, then the flash forward;
# backward = flash backward followed by the INVERSE rotation (negate_sin) of dQ and dK.
def apply_rope_eager_bhsd(x, cos_t, sin_t):
    """Eager interleaved-pair RoPE in fp32 (the reference's fallback / spec, :1451-1468)."""
    cos_b = (cos_t.unsqueeze(0) if cos_t.dim() == 2 else cos_t).unsqueeze(1)
    sin_b = (sin_t.unsqueeze(0) if sin_t.dim() == 2 else sin_t).unsqueeze(1)
    xf = x.float()
    pairs = xf.reshape(*xf.shape[:-1], xf.shape[-1] // 2, 2)
    rotated = torch.stack((-pairs[..., 1], pairs[..., 0]), -1).reshape(xf.shape)
    S = x.shape[2]
    return (xf * cos_b[..., :S, :] + rotated * sin_b[..., :S, :]).to(x.dtype)


class _RopeFlashAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cos_t, sin_t, causal: bool, scale: float):
        q_rot = ops.rope_rotate(q, cos_t, sin_t)
        k_rot = ops.rope_rotate(k, cos_t, sin_t)
        v = v.contiguous()
        o32, lse = ops.attention_forward(q_rot, k_rot, v, scale=scale, causal=causal, out_dtype=torch.float32,
                                         return_lse=True)
        ctx.save_for_backward(q_rot, k_rot, v, o32, lse, cos_t, sin_t)
        ctx.causal, ctx.scale = causal, scale
        return o32.to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        q_rot, k_rot, v, o32, lse, cos_t, sin_t = ctx.saved_tensors
        dq, dk, dv = ops.attention_backward(dout.to(q_rot.dtype).contiguous(), q_rot, k_rot, v, o32, lse,
                                            scale=float(ctx.scale), causal=bool(ctx.causal), keep_fp32=True)
        # RoPE is orthonormal: the gradient w.r.t. the un-rotated tensor is the inverse rotation of the gradient
        dq = ops.rope_rotate(dq, cos_t, sin_t, negate_sin=True)
        dk = ops.rope_rotate(dk, cos_t, sin_t, negate_sin=True)
        return dq.to(q_rot.dtype), dk.to(q_rot.dtype), dv.to(q_rot.dtype), None, None, None, None


def rope_scaled_dot_product_attention(query, key, value, rope_cos, rope_sin, attn_mask=None, is_causal: bool = False,
                                      scale: Optional[float] = None):
    def eager():
        cos_f, sin_f = rope_cos.to(query.device, torch.float32), rope_sin.to(query.device, torch.float32)
        return scaled_dot_product_attention(apply_rope_eager_bhsd(query, cos_f, sin_f),
                                            apply_rope_eager_bhsd(key, cos_f, sin_f), value, attn_mask, 0.0,
                                            is_causal, scale, True)

    needs_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
    ok = (query.is_cuda and key.is_cuda and value.is_cuda and query.dim() == 4 and key.dim() == 4 and value.dim() == 4
          and query.dtype in _SUPPORTED and key.dtype == query.dtype and value.dtype == query.dtype)
    if not ok:
        return eager()
    B, Hq, Sq, D = query.shape
    Hkv, Skv = key.shape[1], key.shape[2]
    if (D % 2 or D > 1024 or key.shape[0] != B or value.shape[:3] != key.shape[:3] or key.shape[3] != D
            or value.shape[3] != D or (Hq != Hkv and (Hkv == 0 or Hq % Hkv)) or Sq != Skv):
        return eager()
    cos_t, sin_t = rope_cos, rope_sin
    if cos_t.dim() == 3 and cos_t.shape[0] == 1:
        cos_t, sin_t = cos_t.squeeze(0), sin_t.squeeze(0)
    cos_t = cos_t.to(query.device, torch.float32).contiguous()
    sin_t = sin_t.to(query.device, torch.float32).contiguous()
    if cos_t.shape != sin_t.shape or cos_t.dim() not in (2, 3) or cos_t.shape[-1] != D or cos_t.shape[-2] != Sq:
        return eager()  # tables longer than Sq are sliced by the eager path only
    if cos_t.dim() == 3 and cos_t.shape[0] != B:
        return eager()
    sm_scale = float(scale) if scale is not None else float(D) ** -0.5
    if needs_grad:
        if attn_mask is not None or Hq != Hkv or D > 256:
            return eager()
        _bump("total")
        _bump("rope_autograd")
        _bump("rope_instream")
        return _RopeFlashAttentionFn.apply(query, key, value, cos_t, sin_t, bool(is_causal), sm_scale)
    q_src = query if query.stride(-1) == 1 else query.contiguous()
    k_src = key if key.stride(-1) == 1 else key.contiguous()
    if attn_mask is None and Hq == Hkv:
        # one call: K rotated once into the stream's workspace, Q rotated inside the attention kernel
        v_src = value if value.stride(-1) == 1 else value.contiguous()
        _bump("total")
        _bump("fp32_instream")   # the same nine counters as the unfused sequence bumps
        _bump("rope_instream")
        return ops.rope_attention_forward(q_src, k_src, v_src, cos_t, sin_t, scale=sm_scale, causal=bool(is_causal))
    q_rot = ops.rope_rotate(q_src, cos_t, sin_t)
    k_rot = ops.rope_rotate(k_src, cos_t, sin_t)
    _bump("rope_instream")
    return scaled_dot_product_attention(q_rot, k_rot, value, attn_mask, 0.0, is_causal, sm_scale, True)
