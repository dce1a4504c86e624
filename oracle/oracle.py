"""numpy front-end of the CPU oracle (oracle/sdpa_ref.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.

Arrays: fp32 -> np.float32, fp16 -> np.float16, bf16 -> np.uint16 holding the
raw bits (numpy has no bfloat16).  Layout is BHSD (metal_sdpa_backend.cpp:188-193).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libsdpa_ref.so"

PREC_FP16, PREC_BF16, PREC_FP32 = 0, 1, 2
MASK_NONE, MASK_BOOL, MASK_ADDITIVE = 0, 1, 2
MSCALAR_BYTE, MSCALAR_FP16, MSCALAR_BF16, MSCALAR_FP32 = 0, 1, 2, 3


def build(force: bool = False) -> Path:
    """Compile libsdpa_ref.so with gcc if it is missing or stale."""
    src = _HERE / "sdpa_ref.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "libsdpa_ref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        i64p = ctypes.POINTER(ctypes.c_int64)
        _lib.ref_sdpa_forward.restype = ctypes.c_int
        _lib.ref_sdpa_forward.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
            ctypes.c_float, ctypes.c_int, ctypes.c_int, i64p, i64p, i64p,
            ctypes.c_void_p, i64p, i64p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int]
        _lib.ref_sdpa_backward.restype = ctypes.c_int
        _lib.ref_sdpa_backward.argtypes = [ctypes.c_void_p] * 10 + [
            ctypes.c_uint32] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        _lib.ref_quantize_symmetric.restype = None
        _lib.ref_quantize_symmetric.argtypes = [
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.ref_pack_int4.restype = None
        _lib.ref_pack_int4.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _lib.ref_unpack_int4.restype = None
        _lib.ref_unpack_int4.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _lib.ref_dequantize.restype = None
        _lib.ref_dequantize.argtypes = [
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        _lib.ref_quantized_forward.restype = ctypes.c_int
        _lib.ref_quantized_forward.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_uint32] * 5 + [
            ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int]
    return _lib


def prec_of(a: np.ndarray) -> int:
    if a.dtype == np.float32:
        return PREC_FP32
    if a.dtype == np.float16:
        return PREC_FP16
    if a.dtype == np.uint16:
        return PREC_BF16
    raise TypeError(f"unsupported operand dtype {a.dtype}")


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bits (MFAFFITests.swift:616-645)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def to_f32(a: np.ndarray) -> np.ndarray:
    return bf16_bits_to_f32(a) if a.dtype == np.uint16 else a.astype(np.float32)


def _i64(arr):
    if arr is None:
        return None
    a = (ctypes.c_int64 * len(arr))(*[int(x) for x in arr])
    return a


def _elem_strides(a: np.ndarray):
    return [s // a.itemsize for s in a.strides]


def sdpa_forward(q, k, v, scale=None, causal=False, mask=None, mask_type=MASK_NONE,
                 return_lse=False):
    """q [B,H,Sq,D], k/v [B,H,Skv,D] (any element strides, last dim contiguous or not).

    mask: numpy array of <=4 dims, bool/uint8 (MASK_BOOL) or fp32/fp16/uint16-bf16
    (MASK_ADDITIVE); broadcast right-aligned onto [B,H,Sq,Skv].
    Returns fp32 O [B,H,Sq,D] (and fp32 LSE [B,H,Sq], natural log).
    """
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    prec = prec_of(q)
    assert prec_of(k) == prec and prec_of(v) == prec
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    out = np.empty((B, H, Sq, D), np.float32)
    lse = np.empty((B, H, Sq), np.float32)
    mptr, mshape, mstr, mnd, mscalar = None, None, None, 0, MSCALAR_BYTE
    if mask is not None and mask_type != MASK_NONE:
        if mask.dtype == np.bool_:
            mask = mask.view(np.uint8)
        mscalar = {np.dtype(np.uint8): MSCALAR_BYTE, np.dtype(np.float16): MSCALAR_FP16,
                   np.dtype(np.uint16): MSCALAR_BF16, np.dtype(np.float32): MSCALAR_FP32}[mask.dtype]
        mptr = mask.ctypes.data
        mshape, mstr, mnd = _i64(mask.shape), _i64(_elem_strides(mask)), mask.ndim
    rc = lib().ref_sdpa_forward(
        q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data, lse.ctypes.data,
        B, H, Sq, Skv, D, float(scale), int(bool(causal)), prec,
        _i64(_elem_strides(q)), _i64(_elem_strides(k)), _i64(_elem_strides(v)),
        mptr, mshape, mstr, mnd, int(mask_type), mscalar)
    if rc != 0:
        raise RuntimeError(f"ref_sdpa_forward rc={rc}")
    return (out, lse) if return_lse else out


def sdpa_backward(dout, q, k, v, out, lse, scale=None, causal=False):
    """Dense contiguous BHSD.  Returns fp32 (dq, dk, dv, D)."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    prec = prec_of(q)
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    dout, q, k, v = (np.ascontiguousarray(a) for a in (dout, q, k, v))
    out = np.ascontiguousarray(out, np.float32)
    lse = np.ascontiguousarray(lse, np.float32)
    dq = np.empty((B, H, Sq, D), np.float32)
    dk = np.empty((B, H, Skv, D), np.float32)
    dv = np.empty((B, H, Skv, D), np.float32)
    dvec = np.empty((B, H, Sq), np.float32)
    rc = lib().ref_sdpa_backward(
        dout.ctypes.data, q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data,
        lse.ctypes.data, dq.ctypes.data, dk.ctypes.data, dv.ctypes.data, dvec.ctypes.data,
        B, H, Sq, Skv, D, float(scale), int(bool(causal)), prec)
    if rc != 0:
        raise RuntimeError(f"ref_sdpa_backward rc={rc}")
    return dq, dk, dv, dvec


def quantize_symmetric(x: np.ndarray, group: int | None = None, bits: int = 8):
    """Returns (int8 values, fp32 scales); QuantizationTests.swift:72-128."""
    x = np.ascontiguousarray(x, np.float32).ravel()
    n = x.size
    group = n if group is None else int(group)
    q = np.empty(n, np.int8)
    scales = np.empty((n + group - 1) // group, np.float32)
    lib().ref_quantize_symmetric(x.ctypes.data, n, group, bits, q.ctypes.data, scales.ctypes.data)
    return q, scales


def pack_int4(q: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, np.int8).ravel()
    out = np.empty((q.size + 1) // 2, np.uint8)
    lib().ref_pack_int4(q.ctypes.data, q.size, out.ctypes.data)
    return out


def unpack_int4(packed: np.ndarray, n: int) -> np.ndarray:
    packed = np.ascontiguousarray(packed, np.uint8).ravel()
    out = np.empty(n, np.int8)
    lib().ref_unpack_int4(packed.ctypes.data, n, out.ctypes.data)
    return out


def dequantize(q: np.ndarray, scales: np.ndarray, group: int | None = None) -> np.ndarray:
    q = np.ascontiguousarray(q, np.int8).ravel()
    scales = np.ascontiguousarray(scales, np.float32)
    group = q.size if group is None else int(group)
    out = np.empty(q.size, np.float32)
    lib().ref_dequantize(q.ctypes.data, q.size, group, scales.ctypes.data, out.ctypes.data)
    return out


def quantized_forward(q, k, v, scale=None, causal=False, mask=None, bits=8, quant_mode=2,
                      block_rows=64):
    """MFABridge+Quantized.swift:227-358 restated: fake-quantise Q,K,V then fp64 SDPA."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    prec = prec_of(q)
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    q, k, v = (np.ascontiguousarray(a) for a in (q, k, v))
    out = np.empty((B, H, Sq, D), np.float32)
    lse = np.empty((B, H, Sq), np.float32)
    mptr = None
    if mask is not None:
        mask = np.ascontiguousarray(np.broadcast_to(mask, (B, H, Sq, Skv)), np.float32)
        mptr = mask.ctypes.data
    rc = lib().ref_quantized_forward(
        q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data, lse.ctypes.data, mptr,
        B, H, Sq, Skv, D, float(scale), int(bool(causal)), bits, quant_mode, block_rows, prec)
    if rc != 0:
        raise RuntimeError(f"ref_quantized_forward rc={rc}")
    return out, lse


def lcg_uniform(count: int, seed: int) -> np.ndarray:
    """The reference tests' deterministic generator (MultiHeadFFITests.swift:1533-1541):
    rng = rng*1664525 + 1013904223 (mod 2^64); value = ((rng % 1e6)/1e6 - 0.5)*2 in fp32."""
    out = np.empty(count, np.float32)
    rng = seed & 0xFFFFFFFFFFFFFFFF
    for i in range(count):
        rng = (rng * 1664525 + 1013904223) & 0xFFFFFFFFFFFFFFFF
        out[i] = (np.float32(rng % 1_000_000) / np.float32(1_000_000.0) - np.float32(0.5)) * np.float32(2.0)
    return out


def rope_rotate(x: np.ndarray, cos: np.ndarray, sin: np.ndarray, negate_sin: bool = False) -> np.ndarray:
    """x [B,H,S,D] (any BHS element strides, contiguous head_dim); cos/sin fp32 [S,D] or [B,S,D].
    Returns the fp32 image of the rotated tensor (MFABridge.swift:269-319)."""
    L = lib()
    L.ref_rope_rotate.restype = None
    L.ref_rope_rotate.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint32] * 4 + [ctypes.c_int64] * 4 + [ctypes.c_int, ctypes.c_int]
    B, H, S, D = x.shape
    cos = np.ascontiguousarray(cos, np.float32)
    sin = np.ascontiguousarray(sin, np.float32)
    tb = S * D if cos.ndim == 3 else 0
    es = _elem_strides(x)
    assert es[3] == 1
    out = np.empty((B, H, S, D), np.float32)
    L.ref_rope_rotate(x.ctypes.data, out.ctypes.data, cos.ctypes.data, sin.ctypes.data, B, H, S, D, es[0], es[1], es[2],
                      tb, int(bool(negate_sin)), prec_of(x))
    return out


def hadamard(x: np.ndarray, block: int) -> np.ndarray:
    """Normalised group-wise FWHT of the flattened array (fp64 math), same shape back."""
    L = lib()
    L.ref_hadamard.restype = None
    L.ref_hadamard.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    v = np.ascontiguousarray(to_f32(x) if x.dtype != np.float64 else x, np.float64).ravel().copy()
    assert v.size % block == 0
    L.ref_hadamard(v.ctypes.data, block, v.size // block)
    return v.reshape(x.shape)


def sdpa_forward_rows(q, k, v, rows, scale=None, causal=False, return_lse=False):
    """The oracle on a subset of query rows against ALL keys: returns O[:, :, rows] (fp32) -- the full-size parity
    checks use it where the whole tensor would take minutes.  `causal` keeps the absolute row indices (top-left
    aligned, as ref_sdpa_forward) by way of a bool mask [len(rows), Skv]."""
    rows = np.asarray(rows, np.int64)
    qs = np.ascontiguousarray(q[:, :, rows])
    mask, mt = None, MASK_NONE
    if causal:
        mask = np.ascontiguousarray(np.arange(k.shape[2])[None, :] <= rows[:, None])
        mt = MASK_BOOL
    return sdpa_forward(qs, k, v, scale=scale, mask=mask, mask_type=mt, return_lse=return_lse)


def round_to(x: np.ndarray, kind: str) -> np.ndarray:
    """fp64/fp32 array rounded to the value set of `kind` ("bf16", "fp16", "fp32"), returned as float64."""
    if kind == "bf16":
        return bf16_bits_to_f32(f32_to_bf16_bits(x.astype(np.float32))).astype(np.float64)
    if kind == "fp16":
        return x.astype(np.float16).astype(np.float64)
    return x.astype(np.float32).astype(np.float64)


def flash_format_floor(q, k, v, rows, kind: str, scale=None, causal=False):
    """What an IDEAL flash kernel with `kind` probabilities gives on these rows: exact fp64 scores and exponentials,
    P rounded once to `kind` (the B operand of the P V MFMA), everything else fp64.  The distance of this from the
    oracle is the part of a 16-bit kernel's error that the operand FORMAT fixes, whatever the kernel does."""
    rows = np.asarray(rows, np.int64)
    qf = to_f32(q).astype(np.float64)[:, :, rows]
    kf, vf = to_f32(k).astype(np.float64), to_f32(v).astype(np.float64)
    if scale is None:
        scale = 1.0 / np.sqrt(q.shape[-1])
    s = np.einsum("bhid,bhjd->bhij", qf, kf) * scale
    if causal:
        s = np.where(np.arange(k.shape[2])[None, :] <= rows[:, None], s, -np.inf)
    p = np.exp(s - s.max(-1, keepdims=True))
    return (np.einsum("bhij,bhjd->bhid", round_to(p, kind), vf) / p.sum(-1, keepdims=True)).astype(np.float32)


def round_e4m3(x: np.ndarray) -> np.ndarray:
    """float array rounded to the OCP fp8 e4m3fn value set (round to nearest even; 3 mantissa bits, normals from 2^-6,
    subnormal step 2^-9, largest finite 448), returned as float64.  |x| must be <= 448 (the kernels guarantee it)."""
    x = np.asarray(x, np.float64)
    a = np.abs(x)
    e = np.clip(np.floor(np.log2(np.maximum(a, 2.0 ** -30))), -6, 8)
    q = 2.0 ** (e - 3)
    return np.sign(x) * np.minimum(np.round(a / q) * q, 448.0)


def fp8_v_tiles(v: np.ndarray, tile: int = 64):
    """The fp8 P V mode's V quantiser restated (fa_quant.hip, quant_mode 3): per (batch, head, 64-key tile) the smallest
    power of two 2^e with absmax / 2^e <= 448, V / 2^e rounded to e4m3.  Returns (de-quantised V as float64, exponents)."""
    vf = to_f32(v).astype(np.float64)
    B, H, S, D = vf.shape
    T = (S + tile - 1) // tile
    pad = np.zeros((B, H, T * tile, D))
    pad[:, :, :S] = vf
    t = pad.reshape(B, H, T, tile, D)
    am = np.abs(t).max(axis=(3, 4))
    e = np.where(am > 0, np.ceil(np.log2(np.maximum(am, 1e-300) / 448.0)), 0.0)
    # frexp convention of the kernel: amax / 448 = f 2^e with f in [0.5, 1): exact powers of two land one exponent higher
    e = np.where((am > 0) & (np.log2(np.maximum(am, 1e-300) / 448.0) == e), e + 1, e)
    sc = 2.0 ** e[..., None, None]
    return (round_e4m3(t / sc) * sc).reshape(B, H, T * tile, D)[:, :, :S], e


def quantized_forward_fp8pv(q, k, v, rows=None, scale=None, causal=False, p_fp8: bool = False, seed=None):
    """quant_mode 3 restated: Q, K fake-quantised block-wise to int8 exactly as quantized_forward does, V through
    fp8_v_tiles, then fp64 SDPA.  p_fp8=False keeps P exact: the reference point the kernel is held against (its own P
    rounding to fp8, whose pattern depends on the deferred reference max, is bounded separately by the test);
    p_fp8=True rounds P (relative to the exact row max) to e4m3 and normalises by the rounded sum -- the statistical
    model of what the kernel adds.  rows: subset of query rows (all keys)."""
    B, H, Sq, D = q.shape
    if scale is None:
        scale = 1.0 / np.sqrt(D)

    def fq(x):
        f = to_f32(x).reshape(B * H, -1)
        out = np.empty_like(f)
        for i in range(B * H):
            qi, sc = quantize_symmetric(f[i], group=64 * D, bits=8)
            out[i] = dequantize(qi, sc, group=64 * D)
        return out.reshape(x.shape).astype(np.float64)

    qf, kf = fq(q), fq(k)
    vf, _ = fp8_v_tiles(v)
    rows = np.arange(Sq) if rows is None else np.asarray(rows)
    s = np.einsum("bhid,bhjd->bhij", qf[:, :, rows], kf) * scale
    if causal:
        s = np.where(np.arange(k.shape[2])[None, :] <= rows[:, None], s, -np.inf)
    p = np.exp(s - s.max(-1, keepdims=True))
    if p_fp8:
        p = round_e4m3(p)
    return (np.einsum("bhij,bhjd->bhid", p, vf) / p.sum(-1, keepdims=True)).astype(np.float32)
