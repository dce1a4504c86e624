"""In-stream launches of the HIP kernels on torch-ROCm tensors (zero-copy).

Counterpart of the reference's try_instream_flash_attention + mps_utils::encode_attention_on_torch_stream
(metal_sdpa_backend.cpp:1308-1446, mps_utils.mm:115-247): raw device pointers, byte offsets folded into
the pointers, BHSD element strides, launched on torch's CURRENT stream -- no host synchronisation.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from umfa._ffi import (MFA_MASK_SCALAR_BF16, MFA_MASK_SCALAR_BYTE, MFA_MASK_SCALAR_FP16, MFA_MASK_SCALAR_FP32,
                       MFA_MASK_TYPE_ADDITIVE, MFA_MASK_TYPE_BOOL, MFA_MASK_TYPE_NONE, MFA_PRECISION_BF16,
                       MFA_PRECISION_FP16, MFA_PRECISION_FP32, MFAError, _check_error, _lib, mfa_context_t)

_PREC = {torch.float16: MFA_PRECISION_FP16, torch.bfloat16: MFA_PRECISION_BF16, torch.float32: MFA_PRECISION_FP32}
_PREC_NAME = {torch.float16: b"fp16", torch.bfloat16: b"bf16", torch.float32: b"fp32"}

_ctx = None


def context():
    """Process-wide context handle (created on first use; metal_sdpa_backend.cpp:936-955)."""
    global _ctx
    if _ctx is None:
        if not torch.cuda.is_available():
            raise RuntimeError("umfa_torch needs a ROCm device: there is no CPU fallback")
        h = mfa_context_t()
        _check_error(_lib.mfa_create_context(ctypes.byref(h)))
        _ctx = h
    return _ctx


def last_kernel() -> str:
    return _lib.umfa_last_kernel_name(context()).decode()


def release_scratch(stream=None, all_streams: bool = False) -> None:
    """Free the library's pooled device scratch of `stream` (default: torch's current stream) or of every stream
    (umfa_release_scratch): only when no graph captured with it will be replayed again."""
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream.cuda_stream
    _check_error(_lib.umfa_release_scratch(context(), ctypes.c_void_p(st), 1 if all_streams else 0))


# launcher switches (include/umfa_abi.h umfa_set_option / umfa_get_option).  The LIBRARY holds the state (seeded once from the
# environment: UMFA_FORCE_W64, UMFA_W64_TAU, ...); this module only reads and writes it.
_OPTION_NAMES = ("softmax_reference", "softmax_tau", "w64_tau", "force_w64", "no_w64", "w64_grid", "w64_skew", "no_mask_flags", "bwd_exact",
                 "bwd_dq", "bwd_persist", "bwd_separate_delta", "no_split", "force_split", "no_dma", "bn64", "pv_fp16", "bwd_ds_store", "no_w64_mask", "ksplit", "no_pipe", "no_w64_mask_lazy", "no_w64_bias", "no_w64_f32_mask", "f32_mask_ratio", "mask_pass_ratio", "no_w64_ragged_mask", "no_mask_realign",
                 "cast_two_pass", "bwd_ds_lab", "cast_u", "quant_block_wg", "cast_wait_us", "cbal", "cbal_delta", "decode_ks", "sync_chunks", "sync_chunked_calls", "mirror_cache_hits")


def get_option(name: str) -> str:
    """The live value of a launcher switch as text."""
    buf = ctypes.create_string_buffer(64)
    _check_error(_lib.umfa_get_option(context(), name.encode(), buf, 64))
    return buf.value.decode()


def pv_fp16_status() -> int:
    """Round 4's status word of the fp16 P V product (1 = a V value left fp16's range).  Always 0 since round 5: V goes into the product as
    V * 2^-e, one power of two per (batch, head) slab chosen on the device, so the condition no longer exists.  Kept for round-4 callers."""
    return int(get_option("pv_fp16_status"))


def set_option(name: str, value) -> None:
    """Context-wide launcher switch, e.g. set_option("softmax_reference", "exact") -- see umfa_set_option in the header."""
    _check_error(_lib.umfa_set_option(context(), name.encode(), str(value).encode()))


class options:
    """with umfa_torch.options(softmax_reference="exact", force_w64=1): ...  -- sets the switches and puts the values the
    library held before back on exit.  Names are validated before anything is applied."""

    def __init__(self, **kw):
        unknown = [k for k in kw if k not in _OPTION_NAMES]
        if unknown:
            raise KeyError(f"unknown launcher switch(es) {unknown}; known: {_OPTION_NAMES}")
        self.kw = kw
        self.prev = {}

    def __enter__(self):
        prev = {k: get_option(k) for k in self.kw}  # everything read before anything is written
        if "w64_tau" in self.kw or "softmax_tau" in self.kw:  # w64_tau also moves the reference policy
            prev.setdefault("softmax_reference", get_option("softmax_reference"))
        for k, v in self.kw.items():
            set_option(k, v)
        self.prev = prev
        return self

    def __exit__(self, *exc):
        ref = self.prev.pop("softmax_reference", None)
        for k, v in self.prev.items():
            set_option("softmax_tau" if k == "w64_tau" else k, v)
        if ref is not None:
            set_option("softmax_reference", ref)
        return False


def _i64(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def _mask_args(mask: Optional[torch.Tensor]):
    if mask is None:
        return None, None, None, 0, MFA_MASK_TYPE_NONE, MFA_MASK_SCALAR_BYTE
    if mask.dtype == torch.bool:
        mt, ms = MFA_MASK_TYPE_BOOL, MFA_MASK_SCALAR_BYTE
    elif mask.dtype == torch.float32:
        mt, ms = MFA_MASK_TYPE_ADDITIVE, MFA_MASK_SCALAR_FP32
    elif mask.dtype == torch.float16:
        mt, ms = MFA_MASK_TYPE_ADDITIVE, MFA_MASK_SCALAR_FP16
    elif mask.dtype == torch.bfloat16:
        mt, ms = MFA_MASK_TYPE_ADDITIVE, MFA_MASK_SCALAR_BF16
    else:
        raise TypeError(f"unsupported mask dtype {mask.dtype}")
    if mask.dim() > 4:
        raise ValueError("attention masks of more than 4 dims are not supported")
    return (ctypes.c_void_p(mask.data_ptr()), _i64(mask.shape), _i64(mask.stride()), mask.dim(), mt, ms)


def _quant_mode(name: str) -> int:
    """"tensor" -> 0, "blockwise" -> 2 (the reference's two modes, MFABridge+Quantized.swift:268-272), "blockwise_fp8pv" -> 3:
    UMFA_QUANT_BLOCKWISE_FP8PV, the MI355X fast mode (block-wise int8 Q K^T, fp8 e4m3 P and V on the 2x-rate MFMA)."""
    n = name.lower()
    return 3 if "fp8" in n else 2 if n.startswith("block") else 0


def attention_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, scale: Optional[float] = None,
                      causal: bool = False, mask: Optional[torch.Tensor] = None, out_dtype=None,
                      return_lse: bool = False, intermediate_dtype=None, out: Optional[torch.Tensor] = None,
                      window=None):
    """q [B,H,Sq,D], k/v [B,H,Skv,D] device tensors (any BHSD strides with a contiguous last dim).

    window=(left, right): sliding-window attention without a mask tensor -- key j attends to row i iff
    i - left <= j <= i + right (add causal=True for a look-back-only window); key tiles outside the band are never
    touched, so the cost follows the band, not Skv.  Exclusive with `mask`.

    out_dtype: torch.float32 (the C-ABI contract) or q.dtype (fused cast-back epilogue).
    Asynchronous on torch's current stream.
    """
    assert q.is_cuda and k.is_cuda and v.is_cuda, "device tensors required"
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    if scale is None:
        scale = D ** -0.5
    out_dtype = out_dtype or q.dtype
    for t in (q, k, v):
        if t.stride(-1) != 1:
            raise ValueError("last dimension must be contiguous")
    if out is None:
        out = torch.empty((B, H, Sq, D), dtype=out_dtype, device=q.device)
    lse = torch.empty((B * H * Sq,), dtype=torch.float32, device=q.device) if return_lse else None
    if window is not None:
        if mask is not None:
            raise ValueError("window and mask are exclusive")
        mptr, mshape, mstr, mnd, mt, ms = None, _i64((int(window[0]), int(window[1]))), None, 0, 3, MFA_MASK_SCALAR_BYTE
    else:
        mptr, mshape, mstr, mnd, mt, ms = _mask_args(mask)
    inter = _PREC[intermediate_dtype or q.dtype]
    stream = torch.cuda.current_stream(q.device).cuda_stream
    _check_error(_lib.umfa_attention_forward_stream(
        context(), ctypes.c_void_p(stream),
        ctypes.c_void_p(q.data_ptr()), _i64(q.stride()), ctypes.c_void_p(k.data_ptr()), _i64(k.stride()),
        ctypes.c_void_p(v.data_ptr()), _i64(v.stride()), ctypes.c_void_p(out.data_ptr()), _PREC[out.dtype],
        ctypes.c_void_p(lse.data_ptr()) if lse is not None else None,
        mptr, mshape, mstr, mnd, mt, ms, B, Sq, Skv, H, D, float(scale), bool(causal), _PREC[q.dtype], inter))
    return (out, lse) if return_lse else out


def attention_encode(q, k, v, out32, *, scale=None, causal=False, mask=None, stream=None):
    """The reference's exact in-stream entry (mfa_attention_encode_mtl): fp32 dense output, precisions as
    strings, byte offsets (0 here: data_ptr() already includes the storage offset)."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    if scale is None:
        scale = D ** -0.5
    mptr, mshape, mstr, mnd, mt, ms = _mask_args(mask)
    s = stream if stream is not None else torch.cuda.current_stream(q.device).cuda_stream
    _check_error(_lib.mfa_attention_encode_mtl(
        context(), ctypes.c_void_p(s),
        ctypes.c_void_p(q.data_ptr()), 0, _i64(q.stride()), ctypes.c_void_p(k.data_ptr()), 0, _i64(k.stride()),
        ctypes.c_void_p(v.data_ptr()), 0, _i64(v.stride()), ctypes.c_void_p(out32.data_ptr()), 0,
        mptr, 0, mshape, mstr, mnd, mt, ms, B, Sq, Skv, H, D, float(scale), bool(causal),
        _PREC_NAME[q.dtype], _PREC_NAME[q.dtype]))
    return out32


class _DevBuf:
    """mfa_buffer_t view of a device tensor (mfa_buffer_from_mtl_buffer: borrowed raw device pointer)."""

    def __init__(self, t: Optional[torch.Tensor]):
        from umfa._ffi import mfa_buffer_t
        self.handle = mfa_buffer_t()
        if t is not None:
            _check_error(_lib.mfa_buffer_from_mtl_buffer(context(), ctypes.c_void_p(t.data_ptr()),
                                                         t.numel() * t.element_size(), ctypes.byref(self.handle)))

    def close(self):
        if self.handle:
            _lib.mfa_destroy_buffer(self.handle)
            self.handle = None


def quantized_attention_forward(q, k, v, *, scale=None, causal=False, mask=None, bits: int = 8,
                                quant_mode: str = "blockwise"):
    """Runtime-quantised SDPA on device tensors through mfa_quantized_forward_with_lse (synchronous, like the
    reference's MetalQuantizedFlashAttentionFn::forward, metal_sdpa_backend.cpp:3142-3267).
    Returns (O fp32 [B,H,Sq,D], LSE fp32 [B*H*Sq]); GPU seconds of quantiser + attention via gpu_latency()."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    if scale is None:
        scale = D ** -0.5
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    out = torch.empty((B, H, Sq, D), dtype=torch.float32, device=q.device)
    lse = torch.empty((B * H * Sq,), dtype=torch.float32, device=q.device)
    m32 = None
    if mask is not None:
        m32 = torch.zeros((B, H, Sq, Skv), dtype=torch.float32, device=q.device)
        m32 = m32.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else m32 + mask.float()
        m32 = m32.contiguous()
    torch.cuda.current_stream(q.device).synchronize()  # the entry runs on the legacy default stream
    bufs = [_DevBuf(t) for t in (q, k, v, out, lse, m32)]
    try:
        _check_error(_lib.mfa_quantized_forward_with_lse(
            context(), *(b.handle for b in bufs), B, Sq, Skv, H, D, float(scale), bool(causal),
            4 if bits == 4 else 3, _quant_mode(quant_mode), _PREC[q.dtype]))
    finally:
        for b in bufs:
            b.close()
    return out, lse


def quantized_attention_forward_stream(q, k, v, *, scale=None, causal=False, mask=None, bits: int = 8,
                                       quant_mode: str = "blockwise", return_lse: bool = False, out=None, lse=None):
    """quantized_attention_forward without the host round trips: umfa_quantized_forward_stream on torch's current
    stream (asynchronous).  Same numbers as the blocking entry.  out / lse: caller-provided fp32 [B,H,Sq,D] / [B*H*Sq]
    (no allocation on the launch path, like attention_forward's out=)."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    if scale is None:
        scale = D ** -0.5
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    if out is None:
        out = torch.empty((B, H, Sq, D), dtype=torch.float32, device=q.device)
    if lse is None:
        lse = torch.empty((B * H * Sq,), dtype=torch.float32, device=q.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and lse.dtype == torch.float32 and lse.numel() == B * H * Sq
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
    tail = (B, Sq, Skv, H, D, float(scale), bool(causal), 4 if bits == 4 else 3, _quant_mode(quant_mode), _PREC[q.dtype])
    if mask is not None and hasattr(_lib, "umfa_quantized_forward_masked_stream"):
        # the mask as the caller has it -- any <= 4-D broadcastable bool / float tensor, read in place with its strides -- instead of the dense
        # fp32 [B, H, Sq, Skv] expansion of the reference's quantised entry (4.3 GB at config 4; a bool [1, 1, S, S] mask is 64 MB)
        mptr, mshape, mstr, mnd, mt, ms = _mask_args(mask)
        _check_error(_lib.umfa_quantized_forward_masked_stream(context(), stream, vp(q), vp(k), vp(v), vp(out), vp(lse), mptr, mshape, mstr, mnd, mt, ms, *tail))
        return (out, lse) if return_lse else out
    m32 = None
    if mask is not None:  # (a library without the masked entry: UMFA_LIBRARY pointing at an older build)
        if mask.dtype == torch.float32 and tuple(mask.shape) == (B, H, Sq, Skv) and mask.is_contiguous():
            m32 = mask
        else:
            m32 = torch.zeros((B, H, Sq, Skv), dtype=torch.float32, device=q.device)
            m32 = (m32.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else m32 + mask.float()).contiguous()
    _check_error(_lib.umfa_quantized_forward_stream(context(), stream, vp(q), vp(k), vp(v), vp(out), vp(lse), vp(m32), *tail))
    return (out, lse) if return_lse else out


def quantized_attention_backward_stream(dout, q, k, v, o32, lse, *, scale=None, causal=False, bits: int = 8,
                                        quant_mode: str = "blockwise"):
    """Backward of quantized_attention_forward_stream, in-stream (umfa_quantized_backward_stream): contiguous BHSD device
    tensors, O fp32 and LSE from the quantised forward.  Returns (dq, dk, dv, status): fp32 gradients and a device
    uint32 that round 4 set when an operand left fp16's range on the 16-bit MFMA engine; every operand enters that engine as
    a power-of-two multiple now (exponents chosen on the device), so it stays 0."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    if scale is None:
        scale = D ** -0.5
    for t in (dout, q, k, v, o32, lse):
        assert t.is_cuda and t.is_contiguous()
    assert o32.dtype == torch.float32 and lse.dtype == torch.float32 and dout.dtype == q.dtype
    dq = torch.empty((B, H, Sq, D), dtype=torch.float32, device=q.device)
    dk = torch.empty((B, H, Skv, D), dtype=torch.float32, device=q.device)
    dv = torch.empty_like(dk)
    status = torch.zeros(1, dtype=torch.int32, device=q.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    _check_error(_lib.umfa_quantized_backward_stream(
        context(), ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream), vp(q), vp(k), vp(v), vp(o32), vp(dout),
        vp(lse), vp(dq), vp(dk), vp(dv), vp(status), B, Sq, Skv, H, D, float(scale), bool(causal), 4 if bits == 4 else 3,
        _quant_mode(quant_mode), _PREC[q.dtype]))
    return dq, dk, dv, status


def attention_backward_gqa(dout, q, k, v, o, lse, *, scale: float, causal: bool = False):
    """dQ [B,Hq,Sq,D], dK, dV [B,Hkv,Skv,D] of grouped-query attention with K / V read IN PLACE (no repeat_interleave
    copies): umfa_attention_backward_gqa_stream, in-stream, gradients in q.dtype.  Returns None when the 16-bit MFMA
    backward cannot serve the call (fp32 operands, head_dim other than 64 / 128 / 256): the caller then expands K / V."""
    B, Hq, Sq, D = q.shape
    Hkv, Skv = k.shape[1], k.shape[2]
    for t in (dout, q, k, v, o, lse):
        assert t.is_cuda and t.is_contiguous()
    if q.dtype == torch.float32 or D not in (64, 128, 256) or Hq % Hkv:
        return None
    assert o.dtype in (torch.float32, q.dtype) and lse.dtype == torch.float32 and dout.dtype == q.dtype
    dq = torch.empty_like(q)
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    dvec = torch.empty((B * Hq * Sq,), dtype=torch.float32, device=q.device)
    rc = _lib.umfa_attention_backward_gqa_stream(
        context(), ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream),
        *(ctypes.c_void_p(t.data_ptr()) for t in (dout, q, k, v, o, lse, dq, dk, dv, dvec)),
        B, Sq, Skv, Hq, Hkv, D, float(scale), bool(causal), _PREC[q.dtype], True, o.dtype != torch.float32)
    if rc == 1:
        return None
    _check_error(rc)
    return dq, dk, dv


def gpu_latency() -> float:
    return float(_lib.mfa_get_gpu_latency(context()))


def attention_backward(dout, q, k, v, o32, lse, *, scale: float, causal: bool = False, grads_in_input_type: bool = True,
                       intermediate_dtype=None, keep_fp32: bool = False):
    """dQ, dK, dV of the SDPA in-stream (umfa_attention_backward_stream): contiguous BHSD device tensors, O fp32 and LSE
    from the forward; asynchronous on torch's current stream.  Gradients come back in q.dtype straight from the kernels
    when the 16-bit MFMA backward serves the call, else fp32 tensors cast afterwards (same values the blocking ABI gives)."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    for t in (dout, q, k, v, o32, lse):
        assert t.is_cuda and t.is_contiguous()
    assert o32.dtype in (torch.float32, q.dtype) and lse.dtype == torch.float32 and dout.dtype == q.dtype
    o_typed = o32.dtype != torch.float32  # O in the operand type (what the forward returned) instead of an fp32 copy
    inter = _PREC[intermediate_dtype or q.dtype]
    dvec = torch.empty((B * H * Sq,), dtype=torch.float32, device=q.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)

    def call(gdt, typed):
        dq = torch.empty((B, H, Sq, D), dtype=gdt, device=q.device)
        dk = torch.empty((B, H, Skv, D), dtype=gdt, device=q.device)
        dv = torch.empty_like(dk)
        rc = _lib.umfa_attention_backward_stream(
            context(), stream, *(ctypes.c_void_p(t.data_ptr()) for t in (dout, q, k, v, o32, lse, dq, dk, dv, dvec)),
            B, Sq, Skv, H, D, float(scale), bool(causal), _PREC[q.dtype], inter, typed, o_typed)
        return rc, dq, dk, dv

    if grads_in_input_type and not keep_fp32 and q.dtype != torch.float32:
        rc, dq, dk, dv = call(q.dtype, True)
        if rc == 0:
            return dq, dk, dv
    rc, dq, dk, dv = call(torch.float32, False)
    _check_error(rc)
    if keep_fp32:  # the caller post-processes the gradients in fp32 (RoPE inverse rotation)
        return dq, dk, dv
    return dq.to(q.dtype), dk.to(q.dtype), dv.to(q.dtype)


def bench_int8(steps: int = 20, warmup: int = 3):
    """int8 block-quantised forward vs the bf16 forward on the same tensors (quantiser pre-pass included),
    for the FLUX shape and BASELINE config 4 (B1 H16 S8192 D128).  GPU time from the library's hipEvents."""
    res = {}
    for name, (B, H, S, D) in {"flux_B1_H24_S4096_D128": (1, 24, 4096, 128), "cfg4_B1_H16_S8192_D128": (1, 16, 8192, 128)}.items():
        torch.manual_seed(0)
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for _ in range(warmup):
            attention_forward(q, k, v, out=out)
        for a, b in ev:
            a.record()
            attention_forward(q, k, v, out=out)
            b.record()
        torch.cuda.synchronize()
        bf = sorted(a.elapsed_time(b) for a, b in ev)
        t8 = []
        for i in range(warmup + steps):
            quantized_attention_forward(q, k, v)
            if i >= warmup:
                t8.append(gpu_latency() * 1e3)
        t8.sort()
        flops = 4.0 * B * H * S * S * D
        res[name] = {"bf16_ms": round(bf[len(bf) // 2], 4), "int8_ms_incl_quantiser": round(t8[len(t8) // 2], 4),
                     "speedup": round(bf[len(bf) // 2] / t8[len(t8) // 2], 3),
                     "int8_TOPs": round(flops / (t8[len(t8) // 2] * 1e-3) / 1e12, 1), "fp32_out": True}
        del q, k, v, out
    return res


def rope_rotate(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, *, negate_sin: bool = False) -> torch.Tensor:
    """Interleaved-pair rotary rotation in-stream (mfa_rope_rotate_encode_mtl, MFABridge.swift:2286-2375).
    x [B,H,S,D] with a contiguous last dim; cos/sin fp32 [S,D] (shared) or [B,S,D], pair-duplicated.
    Returns a dense [B,H,S,D] tensor of x's dtype.  negate_sin=True is the inverse rotation (backward of Q/K)."""
    B, H, S, D = x.shape
    assert x.is_cuda and x.stride(-1) == 1 and cos.dtype == torch.float32 and sin.dtype == torch.float32
    cos, sin = cos.contiguous(), sin.contiguous()
    tb = S * D if cos.dim() == 3 else 0
    out = torch.empty((B, H, S, D), dtype=x.dtype, device=x.device)
    lib = _lib
    lib.mfa_rope_rotate_encode_mtl.restype = ctypes.c_int
    lib.mfa_rope_rotate_encode_mtl.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_bool, ctypes.c_uint32,
                                               ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p]
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _check_error(lib.mfa_rope_rotate_encode_mtl(context(), ctypes.c_void_p(stream), ctypes.c_void_p(x.data_ptr()), 0,
                                                x.stride(0), x.stride(1), x.stride(2), ctypes.c_void_p(out.data_ptr()), 0,
                                                ctypes.c_void_p(cos.data_ptr()), 0, ctypes.c_void_p(sin.data_ptr()), 0,
                                                tb, bool(negate_sin), B, H, S, D, _PREC_NAME[x.dtype]))
    return out


def rope_attention_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, *,
                           scale: Optional[float] = None, causal: bool = False, out_dtype=None,
                           return_lse: bool = False):
    """softmax(rope(q) rope(k)^T * scale) v in ONE in-stream call (umfa_rope_attention_forward_stream): K is rotated
    once into the stream's workspace, Q inside the attention kernel behind its fragment load.  Bit-identical to
    rope_rotate(q), rope_rotate(k), attention_forward (the reference's sequence, metal_sdpa_backend.cpp:1472-1641).
    q, k, v [B,H,S,D] (same S), contiguous last dim; cos / sin fp32 [S,D] or [B,S,D], pair-duplicated."""
    assert q.is_cuda and k.is_cuda and v.is_cuda, "device tensors required"
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    for t in (q, k, v):
        if t.stride(-1) != 1:
            raise ValueError("last dimension must be contiguous")
    if cos.dtype != torch.float32 or sin.dtype != torch.float32 or cos.shape != sin.shape or cos.shape[-2:] != (Sq, D):
        raise ValueError("cos / sin must be fp32 [S, D] or [B, S, D]")
    cos, sin = cos.contiguous(), sin.contiguous()
    tb = Sq * D if cos.dim() == 3 else 0
    if scale is None:
        scale = D ** -0.5
    out = torch.empty((B, H, Sq, D), dtype=out_dtype or q.dtype, device=q.device)
    lse = torch.empty((B * H * Sq,), dtype=torch.float32, device=q.device) if return_lse else None
    stream = torch.cuda.current_stream(q.device).cuda_stream
    _check_error(_lib.umfa_rope_attention_forward_stream(
        context(), ctypes.c_void_p(stream),
        ctypes.c_void_p(q.data_ptr()), _i64(q.stride()), ctypes.c_void_p(k.data_ptr()), _i64(k.stride()),
        ctypes.c_void_p(v.data_ptr()), _i64(v.stride()), ctypes.c_void_p(out.data_ptr()), _PREC[out.dtype],
        ctypes.c_void_p(lse.data_ptr()) if lse is not None else None,
        ctypes.c_void_p(cos.data_ptr()), ctypes.c_void_p(sin.data_ptr()), tb, B, Sq, Skv, H, D, float(scale),
        bool(causal), _PREC[q.dtype], _PREC[q.dtype]))
    return (out, lse) if return_lse else out


def hadamard_rotate(t: torch.Tensor, block_size: int) -> torch.Tensor:
    """In-place group-wise Hadamard rotation of a contiguous fp32 / fp16 tensor (reference: hadamard_rotate_inplace,
    metal_sdpa_backend.cpp:3400-3418).  Applying it twice is the identity."""
    assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float16)
    if t.numel() % block_size:
        raise RuntimeError("Tensor size not divisible by block_size")
    _lib.mfa_hadamard_rotate.restype = ctypes.c_int32
    _lib.mfa_hadamard_rotate.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    context()
    torch.cuda.current_stream(t.device).synchronize()
    buf = _DevBuf(t)
    try:
        _check_error(_lib.mfa_hadamard_rotate(buf.handle, block_size, t.numel() // block_size))
    finally:
        buf.close()
    return t
