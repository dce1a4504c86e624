"""ctypes binding of libMFAFFI.so (MI355X build) -- mirrors the reference's
examples/python-ffi/src/umfa/_ffi.py: same constants, same argtypes (widths per
mfa_ffi.h: bool = 1 byte, head_dim = uint16, scale = float by value), same
MFAError / _check_error helpers.  The library is searched in-tree (../lib), then
$UMFA_LIBRARY, then the system loader.  There is no CPU fallback: if the HIP library is
missing the import fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

MFA_SUCCESS = 0
MFA_ERROR_INVALID_ARGS = 1
MFA_ERROR_MEMORY_ALLOCATION = 2
MFA_ERROR_DEVICE_NOT_SUPPORTED = 3
MFA_ERROR_KERNEL_COMPILATION = 4
MFA_ERROR_EXECUTION_FAILED = 5

MFA_PRECISION_FP16 = 0
MFA_PRECISION_BF16 = 1
MFA_PRECISION_FP32 = 2
MFA_PRECISION_INT8 = 3
MFA_PRECISION_INT4 = 4

MFA_MASK_TYPE_NONE = 0
MFA_MASK_TYPE_BOOL = 1
MFA_MASK_TYPE_ADDITIVE = 2

MFA_MASK_SCALAR_BYTE = 0
MFA_MASK_SCALAR_FP16 = 1
MFA_MASK_SCALAR_BF16 = 2
MFA_MASK_SCALAR_FP32 = 3

mfa_error_t = ctypes.c_int32
mfa_precision_t = ctypes.c_int32
mfa_context_t = ctypes.c_void_p
mfa_buffer_t = ctypes.c_void_p

_ERROR_TEXT = {0: "Success", 1: "Invalid arguments", 2: "Memory allocation failed",
               3: "Device not supported", 4: "Kernel compilation failed", 5: "Execution failed"}


class MFAError(Exception):
    """Raised when an mfa_* call returns a non-zero status."""

    def __init__(self, code: int, message: str = ""):
        self.code = code
        self.message = message or _get_error_string(code)
        super().__init__(f"MFA Error {code}: {self.message}")


def _find_library() -> str:
    here = Path(__file__).resolve().parent
    candidates = [here.parent / "lib" / "libMFAFFI.so"]
    env = os.environ.get("UMFA_LIBRARY")
    if env:
        candidates.insert(0, Path(env))
    candidates += [Path("/usr/local/lib/libMFAFFI.so"), Path("/opt/rocm/lib/libMFAFFI.so")]
    for c in candidates:
        if c.exists():
            return str(c)
    raise RuntimeError(
        "Could not find libMFAFFI.so (the HIP library). Build it with "
        "`make -C universal-metal-flash-attention_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")


_i64p = ctypes.POINTER(ctypes.c_int64)
_u32, _u16, _f32, _b, _i32, _vp, _sz = (ctypes.c_uint32, ctypes.c_uint16, ctypes.c_float, ctypes.c_bool,
                                        ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t)
_DIMS = [_u32, _u32, _u32, _u32, _u16]  # batch, seq_q, seq_kv, heads, head_dim
_MASK_HOST = [_vp, _sz, _i64p, _i64p, _u32, _i32, _i32]


def _load_library(path: str | None = None) -> ctypes.CDLL:
    """path: an explicit library file (tools/ab_inproc.py loads variant builds side by side); default = the search."""
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so, so when torch is
    # installed load it first and let libMFAFFI.so bind to that copy (same SONAME).  Two runtimes
    # in one process cannot share streams or device pointers (and the second one sees no device).
    try:
        import torch  # noqa: F401
    except Exception:  # pure-numpy callers: the system ROCm runtime is used
        pass
    lib = ctypes.CDLL(path or _find_library())

    def sig(name, restype, argtypes):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes

    sig("mfa_create_context", mfa_error_t, [ctypes.POINTER(mfa_context_t)])
    sig("mfa_destroy_context", None, [mfa_context_t])
    sig("mfa_create_buffer", mfa_error_t, [mfa_context_t, _sz, ctypes.POINTER(mfa_buffer_t)])
    sig("mfa_buffer_from_ptr", mfa_error_t, [mfa_context_t, _vp, _sz, ctypes.POINTER(mfa_buffer_t)])
    sig("mfa_buffer_from_ptr_with_strides", mfa_error_t,
        [mfa_context_t, _vp, _sz, _i64p, _i64p, _u32, ctypes.POINTER(mfa_buffer_t)])
    sig("mfa_buffer_from_mtl_buffer", mfa_error_t, [mfa_context_t, _vp, _sz, ctypes.POINTER(mfa_buffer_t)])
    sig("mfa_buffer_from_mtl_buffer_with_strides", mfa_error_t,
        [mfa_context_t, _vp, _sz, _i64p, _i64p, _u32, ctypes.POINTER(mfa_buffer_t)])
    sig("mfa_buffer_contents", _vp, [mfa_buffer_t])
    sig("mfa_destroy_buffer", None, [mfa_buffer_t])

    sig("mfa_attention_forward", mfa_error_t,
        [mfa_context_t] + [mfa_buffer_t] * 4 + _DIMS + [_f32, _b] + [mfa_precision_t] * 3 + [_b] * 4 + _MASK_HOST)
    sig("mfa_attention_forward_str", mfa_error_t,
        [mfa_context_t] + [mfa_buffer_t] * 4 + _DIMS + [_f32, _b] + [ctypes.c_char_p] * 3 + [_b] * 4 + _MASK_HOST)
    sig("mfa_attention_encode_mtl", mfa_error_t,
        [mfa_context_t, _vp,
         _vp, ctypes.c_int64, _i64p, _vp, ctypes.c_int64, _i64p, _vp, ctypes.c_int64, _i64p,
         _vp, ctypes.c_int64, _vp, ctypes.c_int64, _i64p, _i64p, _u32, _i32, _i32] + _DIMS +
        [_f32, _b, ctypes.c_char_p, ctypes.c_char_p])
    sig("umfa_attention_forward_stream", mfa_error_t,
        [mfa_context_t, _vp, _vp, _i64p, _vp, _i64p, _vp, _i64p, _vp, _i32, _vp,
         _vp, _i64p, _i64p, _u32, _i32, _i32] + _DIMS + [_f32, _b, _i32, _i32])
    sig("umfa_rope_attention_forward_stream", mfa_error_t,
        [mfa_context_t, _vp, _vp, _i64p, _vp, _i64p, _vp, _i64p, _vp, _i32, _vp, _vp, _vp, ctypes.c_int64] + _DIMS +
        [_f32, _b, _i32, _i32])
    sig("mfa_attention_forward_with_lse", _i32,
        [mfa_context_t] + [mfa_buffer_t] * 5 + _DIMS + [_f32, _b, _i32, _i32] + [_b] * 4)
    sig("mfa_attention_backward", mfa_error_t,
        [mfa_context_t] + [mfa_buffer_t] * 10 + _DIMS + [_f32, _b, mfa_precision_t, mfa_precision_t] + [_b] * 4)
    sig("mfa_quantized_forward_with_lse", _i32,
        [mfa_context_t] + [mfa_buffer_t] * 6 + _DIMS + [_f32, _b, _i32, _i32, _i32])
    sig("mfa_quantized_backward", _i32,
        [mfa_context_t] + [mfa_buffer_t] * 10 + _DIMS + [_f32, _b, _i32, _i32, _i32])
    _legacy = ([mfa_context_t] + [mfa_buffer_t] * 4 + _DIMS + [_f32, _b] + [_f32, _i32] * 3)
    sig("mfa_attention_forward_quantized", mfa_error_t, _legacy + [mfa_precision_t] * 4 + [_b] * 4)
    sig("mfa_attention_forward_quantized_direct", mfa_error_t, _legacy + [_i32] * 4 + [_b] * 4)
    sig("mfa_multihead_attention_quantized_direct", mfa_error_t, _legacy + [_i32] * 3)
    sig("mfa_attention_forward_quantized_unified", mfa_error_t,
        _legacy + [mfa_precision_t] * 4 + [_i32, _u32, _u32, _u32, _b, _b] + [_b] * 4)
    sig("mfa_attention_forward_quantized_enhanced", mfa_error_t,
        _legacy + [mfa_precision_t] * 4 + [_i32, _u32, _u32, _u32, _b, _b] + [_b] * 4)
    # pre-quantised backward (mfa_ffi.h:480-624): 8 buffers, dims, 3 x (scale, zero point), 3 precisions, 5 bools
    _qb_tail = [_f32, _i32] * 3 + [_i32] * 3 + [_b] * 5
    _qb_blocks = [mfa_buffer_t] * 6 + [_u32] * 4
    _u16 = ctypes.c_uint16
    sig("mfa_attention_backward_query_quantized", _i32, [mfa_context_t] + [mfa_buffer_t] * 8 + [_u32] * 4 + [_u16] + _qb_tail)
    sig("mfa_attention_backward_kv_quantized", _i32, [mfa_context_t] + [mfa_buffer_t] * 8 + [_u32] * 4 + [_u16] + _qb_tail)
    sig("mfa_attention_backward_query_quantized_ex", _i32,
        [mfa_context_t] + [mfa_buffer_t] * 8 + [_u32] * 5 + [_u16] + _qb_tail + _qb_blocks)
    sig("mfa_attention_backward_kv_quantized_ex", _i32,
        [mfa_context_t] + [mfa_buffer_t] * 8 + [_u32] * 5 + [_u16] + _qb_tail + _qb_blocks)
    sig("mfa_set_scale_arrays", mfa_error_t,
        [mfa_context_t, ctypes.POINTER(_f32), _u32, ctypes.POINTER(_f32), _u32, ctypes.POINTER(_f32), _u32])

    sig("mfa_error_string", _vp, [mfa_error_t])  # strdup'd: we free() it ourselves
    sig("mfa_is_device_supported", _b, [])
    sig("mfa_get_version", None, [ctypes.POINTER(ctypes.c_int)] * 3)
    sig("mfa_get_gpu_latency", ctypes.c_double, [mfa_context_t])
    sig("mfa_has_native_bfloat", _i32, [])
    sig("mfa_has_native_bfloat_msl32", _i32, [])
    sig("mfa_get_quantized_layout", None, [ctypes.c_int, _vp])
    sig("mfa_get_quantized_capabilities", None, [_vp])
    sig("umfa_attention_backward_stream", mfa_error_t,
        [mfa_context_t, _vp] + [_vp] * 10 + _DIMS + [_f32, _b, _i32, _i32, _b, _b])
    sig("umfa_quantized_forward_stream", mfa_error_t,
        [mfa_context_t, _vp] + [_vp] * 6 + _DIMS + [_f32, _b, _i32, _i32, _i32])
    if path is None or hasattr(lib, "umfa_quantized_forward_masked_stream"):
        sig("umfa_quantized_forward_masked_stream", mfa_error_t,
            [mfa_context_t, _vp] + [_vp] * 6 + [_vp, _vp, _u32, _i32, _i32] + _DIMS + [_f32, _b, _i32, _i32, _i32])
    if path is None or hasattr(lib, "umfa_attention_backward_gqa_stream"):
        sig("umfa_attention_backward_gqa_stream", mfa_error_t,
            [mfa_context_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, ctypes.c_uint16,
             ctypes.c_float, ctypes.c_bool, _i32, ctypes.c_bool, ctypes.c_bool])
    if path is None or hasattr(lib, "umfa_quantized_backward_stream"):
        sig("umfa_quantized_backward_stream", mfa_error_t,
            [mfa_context_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, ctypes.c_uint16,
             ctypes.c_float, ctypes.c_bool, _i32, _i32, _i32])
    sig("umfa_last_kernel_name", ctypes.c_char_p, [mfa_context_t])
    if path is None or hasattr(lib, "umfa_release_scratch"):
        sig("umfa_release_scratch", mfa_error_t, [mfa_context_t, _vp, _i32])
    if path is None or hasattr(lib, "umfa_set_option"):  # (tools/ab_inproc.py also loads older builds by explicit path)
        sig("umfa_set_option", _i32, [mfa_context_t, ctypes.c_char_p, ctypes.c_char_p])
    if path is None or hasattr(lib, "umfa_get_option"):
        sig("umfa_get_option", _i32, [mfa_context_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t])
    sig("umfa_quantize_rows", _i32, [mfa_context_t, _vp, _vp, _i32, _u32, _u32, _u32, _i32, _i32, _vp, _vp,
                                     ctypes.POINTER(_u32)])
    return lib


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]
_libc.free.restype = None


def _get_error_string(code: int) -> str:
    try:
        p = _lib.mfa_error_string(code)
        if p:
            text = ctypes.string_at(p).decode("utf-8")
            _libc.free(p)  # caller frees (mfa_ffi.h:448-450)
            return text
    except Exception:
        pass
    return _ERROR_TEXT.get(code, f"Unknown error code: {code}")


def _check_error(code: int) -> None:
    if code != MFA_SUCCESS:
        raise MFAError(code)


_lib = _load_library()
