"""High-level numpy API over the C ABI -- the MI355X counterpart of the reference's
examples/python-ffi/src/umfa/core.py (same public names, argument meaning and errors).

Differences that follow from the boundary contract:
  * attention outputs are fp32 at the ABI (MFABridge.swift:1089 ignores output_precision), so the
    output buffer is allocated as fp32 and cast back to q's dtype -- the reference allocates
    `np.zeros_like(q)` (core.py:370), which under-allocates for fp16 inputs;
  * host arrays are mirrored in HBM by the library (discrete GPU); torch-ROCm tensors should use
    umfa_torch, which passes device pointers and is zero-copy;
  * `layout="bhsd"` (extension) lifts the reference's H == 1 restriction for 4-D arrays.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import NamedTuple, Optional, Tuple, Union

import numpy as np

from ._ffi import (MFA_MASK_SCALAR_BF16, MFA_MASK_SCALAR_BYTE, MFA_MASK_SCALAR_FP16, MFA_MASK_SCALAR_FP32,
                   MFA_MASK_TYPE_ADDITIVE, MFA_MASK_TYPE_BOOL, MFA_MASK_TYPE_NONE, MFA_PRECISION_BF16,
                   MFA_PRECISION_FP16, MFA_PRECISION_FP32, MFA_PRECISION_INT4, MFA_PRECISION_INT8, MFAError,
                   _check_error, _lib, mfa_buffer_t, mfa_context_t)

Precision = Union[str, int]
FloatArray = np.ndarray


class MFAContext:
    """Owns one reference on the process-wide MFA context (MFABridge.swift:782-805)."""

    def __init__(self):
        handle = mfa_context_t()
        _check_error(_lib.mfa_create_context(ctypes.byref(handle)))
        self._handle = handle
        self._finalizer = weakref.finalize(self, MFAContext._cleanup, handle)

    def __enter__(self) -> "MFAContext":
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def close(self):
        if self._handle:
            self._finalizer.detach()
            MFAContext._cleanup(self._handle)
            self._handle = mfa_context_t()

    @staticmethod
    def _cleanup(handle):
        if handle:
            _lib.mfa_destroy_context(handle)

    @property
    def handle(self):
        return self._handle

    @property
    def gpu_latency(self) -> float:
        """Seconds of GPU time of the last synchronous op (mfa_get_gpu_latency)."""
        return float(_lib.mfa_get_gpu_latency(self._handle))

    @property
    def last_kernel(self) -> str:
        return _lib.umfa_last_kernel_name(self._handle).decode()

    def __bool__(self) -> bool:
        return bool(self._handle)


class MFABuffer:
    """Wraps a numpy array (zero-copy at the ABI: mfa_buffer_from_ptr) or allocates `size` bytes."""

    def __init__(self, context: MFAContext, data: Optional[np.ndarray] = None, size: Optional[int] = None):
        self._context = context
        self._array = data
        handle = mfa_buffer_t()
        if data is not None:
            if not data.flags.c_contiguous:
                raise ValueError("Array must be C-contiguous for zero-copy")
            _check_error(_lib.mfa_buffer_from_ptr(context.handle, ctypes.c_void_p(data.ctypes.data), data.nbytes,
                                                  ctypes.byref(handle)))
        elif size is not None:
            _check_error(_lib.mfa_create_buffer(context.handle, size, ctypes.byref(handle)))
        else:
            raise ValueError("Must provide either data array or buffer size")
        self._handle = handle
        self._finalizer = weakref.finalize(self, MFABuffer._cleanup, handle)

    def close(self):
        if self._handle:
            self._finalizer.detach()
            MFABuffer._cleanup(self._handle)
            self._handle = mfa_buffer_t()

    @staticmethod
    def _cleanup(handle):
        if handle:
            _lib.mfa_destroy_buffer(handle)

    @property
    def handle(self):
        return self._handle

    def contents_ptr(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(_lib.mfa_buffer_contents(self._handle))

    def __bool__(self) -> bool:
        return bool(self._handle)


_PRECISIONS = {"fp16": MFA_PRECISION_FP16, "half": MFA_PRECISION_FP16, "float16": MFA_PRECISION_FP16,
               "bf16": MFA_PRECISION_BF16, "bfloat16": MFA_PRECISION_BF16, "fp32": MFA_PRECISION_FP32,
               "float": MFA_PRECISION_FP32, "float32": MFA_PRECISION_FP32, "int8": MFA_PRECISION_INT8,
               "int4": MFA_PRECISION_INT4}


def _parse_precision(precision: Precision) -> int:
    if isinstance(precision, (int, np.integer)):
        return int(precision)
    key = str(precision).lower()
    if key not in _PRECISIONS:
        raise ValueError(f"Unknown precision: {precision}. Use one of {list(_PRECISIONS.keys())}")
    return _PRECISIONS[key]


class _MaskMetadata(NamedTuple):
    array: np.ndarray
    ptr: ctypes.c_void_p
    size_bytes: int
    shape: ctypes.Array
    strides: ctypes.Array
    ndim: int
    mask_type: int
    mask_scalar: int


def _prepare_mask_metadata(mask, target_shape: Tuple[int, ...], bf16_bits: bool = False) -> _MaskMetadata:
    """Mask -> contiguous array + FFI metadata.  Unlike the reference (core.py:206-265) the mask is NOT
    expanded to the full target shape: broadcast dims keep size 1 and the kernel broadcasts them."""
    arr = np.asarray(mask)
    mtype, mscalar = MFA_MASK_TYPE_ADDITIVE, MFA_MASK_SCALAR_FP32
    if arr.dtype == np.bool_:
        mtype, mscalar = MFA_MASK_TYPE_BOOL, MFA_MASK_SCALAR_BYTE
    elif arr.dtype == np.float16:
        mscalar = MFA_MASK_SCALAR_FP16
    elif arr.dtype == np.uint16 and bf16_bits:
        mscalar = MFA_MASK_SCALAR_BF16
    elif arr.dtype == np.float32:
        mscalar = MFA_MASK_SCALAR_FP32
    elif arr.dtype == np.float64:
        arr = arr.astype(np.float32)
    elif arr.dtype in (np.int8, np.uint8, np.int16, np.uint16):
        mtype, mscalar = MFA_MASK_TYPE_BOOL, MFA_MASK_SCALAR_BYTE
        arr = arr.astype(np.bool_)
    else:
        raise ValueError("Unsupported attention mask dtype. Use bool for binary masks or "
                         "float16/bfloat16/float32 values for additive masks.")
    try:
        np.broadcast_shapes(arr.shape, target_shape)
        if arr.ndim > len(target_shape):
            raise ValueError
    except ValueError as exc:
        raise ValueError(f"Attention mask with shape {arr.shape} cannot broadcast to {target_shape}.") from exc
    view = np.ascontiguousarray(arr)
    shape = [int(d) for d in view.shape]
    strides = [int(s // view.itemsize) for s in view.strides]
    return _MaskMetadata(view, ctypes.c_void_p(view.ctypes.data), view.nbytes,
                         (ctypes.c_int64 * len(shape))(*shape), (ctypes.c_int64 * len(strides))(*strides),
                         len(shape), mtype, mscalar)


def _dims(q, k, v, layout: str):
    if q.ndim == 2:
        sq, d = q.shape
        skv = k.shape[0]
        if k.shape != (skv, d) or v.shape != (skv, d):
            raise ValueError(f"Shape mismatch: q={q.shape}, k={k.shape}, v={v.shape}")
        return 1, sq, skv, 1, d
    if q.ndim == 4:
        if layout == "bhsd":
            b, h, sq, d = q.shape
            skv = k.shape[2]
            if k.shape != (b, h, skv, d) or v.shape != (b, h, skv, d):
                raise ValueError(f"Shape mismatch: q={q.shape}, k={k.shape}, v={v.shape}")
            return b, sq, skv, h, d
        b, sq, h, d = q.shape
        skv = k.shape[1]
        if k.shape != (b, skv, h, d) or v.shape != (b, skv, h, d):
            raise ValueError(f"Shape mismatch: q={q.shape}, k={k.shape}, v={v.shape}")
        if h != 1:  # reference core.py:336-340
            raise ValueError("Multi-head attention not yet supported. Use num_heads=1 or 2D arrays.")
        return b, sq, skv, h, d
    raise ValueError(f"Invalid tensor dimensions. Expected 2D or 4D, got q.shape={q.shape}")


def _elem_bytes(prec: int) -> int:
    return 4 if prec == MFA_PRECISION_FP32 else 2


def flash_attention_forward(context: MFAContext, q: FloatArray, k: FloatArray, v: FloatArray, *,
                            attn_mask=None, causal: bool = False, softmax_scale: Optional[float] = None,
                            input_precision: Precision = "fp16", intermediate_precision: Precision = "fp16",
                            output_precision: Precision = "fp16", layout: str = "bshd",
                            return_lse: bool = False):
    """softmax(scale q k^T [+causal] [+mask]) v through mfa_attention_forward.

    q: [seq_q, head_dim] or [batch, seq_q, heads(=1), head_dim] (`layout="bhsd"`: [batch, heads, seq, head_dim]).
    bf16 operands are passed as uint16 bit patterns with input_precision="bf16".
    attn_mask: bool (True/nonzero = attend, the kernel's convention -- MFABridge.swift:201-205) or additive
    float mask broadcastable to [batch, heads, seq_q, seq_kv].
    Returns an array shaped like q in q's dtype (fp32 for bf16-bit inputs); with return_lse also the fp32
    log-sum-exp [batch*heads*seq_q].
    """
    if not all(isinstance(x, np.ndarray) for x in (q, k, v)):
        raise TypeError("q, k, v must be numpy arrays")
    b, sq, skv, h, d = _dims(q, k, v, layout)
    if softmax_scale is None:
        softmax_scale = 1.0 / np.sqrt(d)
    in_prec = _parse_precision(input_precision)
    inter_prec = _parse_precision(intermediate_precision)
    out_prec = _parse_precision(output_precision)
    if q.itemsize != _elem_bytes(in_prec):
        raise ValueError(f"input_precision={input_precision} does not match array dtype {q.dtype}")

    mask_meta = None
    if attn_mask is not None:
        if return_lse:
            raise ValueError("mfa_attention_forward_with_lse takes no mask (MFABridge.swift:3078-3166)")
        target = (sq, skv) if q.ndim == 2 else (b, h, sq, skv)
        mask_meta = _prepare_mask_metadata(attn_mask, target, bf16_bits=(in_prec == MFA_PRECISION_BF16))

    out32 = np.zeros(q.shape, np.float32)
    lse = np.zeros(b * h * sq, np.float32) if return_lse else None
    bufs = [MFABuffer(context, a) for a in (q, k, v, out32)]
    if return_lse:
        bufs.append(MFABuffer(context, lse))
    try:
        if return_lse:
            _check_error(_lib.mfa_attention_forward_with_lse(
                context.handle, *(x.handle for x in bufs), b, sq, skv, h, d, softmax_scale, causal, in_prec,
                inter_prec, False, False, False, False))
        else:
            _check_error(_lib.mfa_attention_forward(
                context.handle, *(x.handle for x in bufs), b, sq, skv, h, d, softmax_scale, causal, in_prec,
                inter_prec, out_prec, False, False, False, False,
                mask_meta.ptr if mask_meta else None, mask_meta.size_bytes if mask_meta else 0,
                mask_meta.shape if mask_meta else None, mask_meta.strides if mask_meta else None,
                mask_meta.ndim if mask_meta else 0, mask_meta.mask_type if mask_meta else MFA_MASK_TYPE_NONE,
                mask_meta.mask_scalar if mask_meta else MFA_MASK_SCALAR_BYTE))
    finally:
        for x in bufs:
            x.close()
    output = out32 if q.dtype == np.uint16 else out32.astype(q.dtype, copy=False)
    return (output, lse) if return_lse else output


def attention(q: FloatArray, k: FloatArray, v: FloatArray, *, attn_mask=None, causal: bool = False,
              softmax_scale: Optional[float] = None, precision: Precision = "fp16", **kw) -> FloatArray:
    """Convenience wrapper that creates a context per call (reference core.py:420-449)."""
    with MFAContext() as ctx:
        return flash_attention_forward(ctx, q, k, v, attn_mask=attn_mask, causal=causal,
                                       softmax_scale=softmax_scale, input_precision=precision,
                                       intermediate_precision=precision, output_precision=precision, **kw)


def quantized_attention(context: MFAContext, q: FloatArray, k: FloatArray, v: FloatArray, *,
                        causal: bool = False, softmax_scale: Optional[float] = None,
                        precision: Precision = "int8", quant_mode: str = "blockwise",
                        input_precision: Optional[Precision] = None, layout: str = "bshd",
                        attn_mask: Optional[np.ndarray] = None, return_lse: bool = False):
    """Runtime-quantised attention (Q, K and V symmetric INT8/INT4) through
    mfa_quantized_forward_with_lse (MFABridge+Quantized.swift:227-358).

    The reference's `quantized_attention` (core.py:452-630) quantises on the host with numpy and calls
    the legacy mfa_attention_forward_quantized, which ignores the quantisation arguments; this one uses
    the entry point that actually quantises, on the GPU.
    """
    if not all(isinstance(x, np.ndarray) for x in (q, k, v)):
        raise TypeError("q, k, v must be numpy arrays")
    b, sq, skv, h, d = _dims(q, k, v, layout)
    if softmax_scale is None:
        softmax_scale = 1.0 / np.sqrt(d)
    target = _parse_precision(precision)
    if target not in (MFA_PRECISION_INT8, MFA_PRECISION_INT4):
        raise ValueError("precision must be 'int8' or 'int4'")
    if input_precision is None:
        input_precision = {np.dtype(np.float32): "fp32", np.dtype(np.float16): "fp16",
                           np.dtype(np.uint16): "bf16"}[q.dtype]
    in_prec = _parse_precision(input_precision)
    mode = {"tensor": 0, "tensorwise": 0, "tensor_wise": 0, "blockwise": 2, "block_wise": 2}[quant_mode]
    out32 = np.zeros(q.shape, np.float32)
    lse = np.zeros(b * h * sq, np.float32)
    arrays = [q, k, v, out32, lse]
    mask32 = None
    if attn_mask is not None:
        mask32 = np.ascontiguousarray(np.broadcast_to(np.asarray(attn_mask, np.float32), (b, h, sq, skv)))
        arrays.append(mask32)
    bufs = [MFABuffer(context, a) for a in arrays]
    try:
        _check_error(_lib.mfa_quantized_forward_with_lse(
            context.handle, *(x.handle for x in bufs[:5]), bufs[5].handle if mask32 is not None else None,
            b, sq, skv, h, d, softmax_scale, causal, target, mode, in_prec))
    finally:
        for x in bufs:
            x.close()
    output = out32 if q.dtype == np.uint16 else out32.astype(q.dtype, copy=False)
    return (output, lse) if return_lse else output


def attention_backward(context: MFAContext, dout, q, k, v, out, lse, *, causal: bool = False,
                       softmax_scale: Optional[float] = None, input_precision: Precision = "fp32",
                       intermediate_precision: Optional[Precision] = None, layout: str = "bhsd"):
    """dQ, dK, dV (fp32) through mfa_attention_backward (MFABridge.swift:3171-3282)."""
    b, sq, skv, h, d = _dims(q, k, v, layout)
    if softmax_scale is None:
        softmax_scale = 1.0 / np.sqrt(d)
    in_prec = _parse_precision(input_precision)
    inter = in_prec if intermediate_precision is None else _parse_precision(intermediate_precision)
    out = np.ascontiguousarray(out, np.float32)
    lse = np.ascontiguousarray(lse, np.float32)
    dq = np.zeros(q.shape, np.float32)
    dk = np.zeros(k.shape, np.float32)
    dv = np.zeros(v.shape, np.float32)
    dvec = np.zeros(b * h * sq, np.float32)
    bufs = [MFABuffer(context, a) for a in (dout, q, k, v, out, lse, dq, dk, dv, dvec)]
    try:
        _check_error(_lib.mfa_attention_backward(context.handle, *(x.handle for x in bufs), b, sq, skv, h, d,
                                                 softmax_scale, causal, in_prec, inter, False, False, False, False))
    finally:
        for x in bufs:
            x.close()
    return dq, dk, dv, dvec


def prequantized_backward(context: MFAContext, q, k, v, out, dout, lse, *, q_scale=1.0, k_scale=1.0, v_scale=1.0,
                          q_zero_point=0, k_zero_point=0, v_zero_point=0, q_precision="int8", k_precision="int8",
                          v_precision="int8", causal: bool = False, num_heads: Optional[int] = None,
                          num_kv_heads: Optional[int] = None, head_dim: Optional[int] = None, seq_len_q=None,
                          seq_len_kv=None, batch_size: int = 1, q_block_scales=None, k_block_scales=None,
                          v_block_scales=None, q_block_size: int = 0, k_block_size: int = 0, v_block_size: int = 0):
    """dQ, dK, dV (fp32) for caller-quantised operands through mfa_attention_backward_query_quantized_ex + _kv_ (the
    query call produces D, the kv call consumes it; MFABridge.swift:1699-2163).  q/k/v: raw quantised arrays (int8, or
    uint8 holding packed int4) laid out [B, H, S, D]; shapes are passed explicitly because packed int4 has none."""
    prec = {"fp16": 0, "float16": 0, "bf16": 1, "bfloat16": 1, "fp32": 2, "float32": 2, "int8": 3, "int4": 4}
    B, H, Hkv, Sq, Skv, D = batch_size, num_heads, num_kv_heads or num_heads, seq_len_q, seq_len_kv, head_dim
    out = np.ascontiguousarray(out, np.float32)
    dout = np.ascontiguousarray(dout, np.float32)
    lse = np.ascontiguousarray(lse, np.float32)
    dq = np.zeros((B, H, Sq, D), np.float32)
    dk = np.zeros((B, Hkv, Skv, D), np.float32)
    dv = np.zeros((B, Hkv, Skv, D), np.float32)
    dvec = np.zeros(B * H * Sq, np.float32)

    def opt(a):
        return MFABuffer(context, np.ascontiguousarray(a, np.float32)) if a is not None else None
    bufs = [MFABuffer(context, np.ascontiguousarray(a)) for a in (q, k, v, out, dout, lse, dq, dk, dv, dvec)]
    blk = [opt(q_block_scales), opt(k_block_scales), opt(v_block_scales)]
    bq, bk, bv, bo, bdo, bl, bdq, bdk, bdv, bd = (x.handle for x in bufs)
    hb = [x.handle if x is not None else None for x in blk]
    tail = (float(q_scale), int(q_zero_point), float(k_scale), int(k_zero_point), float(v_scale), int(v_zero_point),
            prec[q_precision], prec[k_precision], prec[v_precision], bool(causal), False, False, False, False)
    blocks = (hb[0], None, hb[1], None, hb[2], None, int(q_block_size), int(k_block_size), int(v_block_size), 0)
    try:
        _check_error(_lib.mfa_attention_backward_query_quantized_ex(context.handle, bq, bk, bv, bo, bdo, bl, bdq, bd, B, Sq,
                                                                     Skv, H, Hkv, D, *tail, *blocks))
        _check_error(_lib.mfa_attention_backward_kv_quantized_ex(context.handle, bq, bk, bv, bdo, bl, bd, bdk, bdv, B, Sq,
                                                                  Skv, H, Hkv, D, *tail, *blocks))
    finally:
        for x in bufs + [y for y in blk if y is not None]:
            x.close()
    return dq, dk, dv, dvec
