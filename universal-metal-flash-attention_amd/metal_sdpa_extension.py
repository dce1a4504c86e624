"""metal_sdpa_extension -- ROCm drop-in for the reference's pybind11 module of the same name
(examples/pytorch-custom-op-ffi/src/python_bindings.cpp:40-426), so existing callers
(`import metal_sdpa_extension as ext; ext.register_backend(); ext.metal_scaled_dot_product_attention(...)`)
run unchanged on MI355X.  Same function names, argument names and defaults; everything routes through the C ABI of
libMFAFFI.so via `umfa_torch` (no CPU fallback: importing this module without the library fails).

Covered: backend registration, the SDPA / RoPE-SDPA / autograd / quantised entry points, quantisation mode + dispatch
counters, Hadamard rotation, the QUANT_* constants, the enums and plain-data config classes, the output-precision
helpers and capability queries.  Not built (they raise NotImplementedError with the ABI's error-3 wording, like the
C symbols return 3): the MLA context and the sparse indexer -- different operators, out of the attention path's scope.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Optional

import torch

import umfa_torch
from umfa import _ffi
from umfa_torch import ops as _ops
from umfa_torch import sdpa as _sdpa

QUANT_NONE = _sdpa.QUANT_NONE
QUANT_INT8 = _sdpa.QUANT_INT8
QUANT_INT4 = _sdpa.QUANT_INT4
QUANT_TENSOR_WISE = _sdpa.QUANT_TENSOR_WISE
QUANT_BLOCK_WISE = _sdpa.QUANT_BLOCK_WISE


class QuantizationPrecision(enum.IntEnum):  # python_bindings.cpp:177-182 (values = mfa_precision_t)
    FP16 = 0
    BF16 = 1
    FP32 = 2
    INT8 = 3
    INT4 = 4


class QuantizationGranularity(enum.IntEnum):  # :185-189
    TENSOR_WISE = 0
    ROW_WISE = 1
    BLOCK_WISE = 2
    HYBRID = 3


class HybridStrategy(enum.IntEnum):  # :192-195
    PERFORMANCE_FIRST = 0
    ACCURACY_FIRST = 1
    BALANCED = 2


class OutputPrecision(enum.IntEnum):  # :198-201
    FP32 = 0
    FP16 = 1
    BF16 = 2


_OUT_DTYPE = {OutputPrecision.FP32: torch.float32, OutputPrecision.FP16: torch.float16, OutputPrecision.BF16: torch.bfloat16}


@dataclass
class BlockSizeConfig:  # :204-211
    query_block_size: int = 64
    key_block_size: int = 64
    value_block_size: int = 64
    head_block_size: int = 0


@dataclass
class TensorAnalysisMetrics:  # :214-223
    dynamic_range: float = 0.0
    variance: float = 0.0
    mean_abs_value: float = 0.0
    tensor_size: int = 0
    memory_footprint: int = 0
    sparsity_ratio: float = 0.0
    has_outliers: bool = False
    quantization_error_estimate: float = 0.0


@dataclass
class HybridGranularityConfig:  # :226-234
    query_granularity: QuantizationGranularity = QuantizationGranularity.BLOCK_WISE
    key_granularity: QuantizationGranularity = QuantizationGranularity.BLOCK_WISE
    value_granularity: QuantizationGranularity = QuantizationGranularity.BLOCK_WISE
    query_blocks: BlockSizeConfig = field(default_factory=BlockSizeConfig)
    key_blocks: BlockSizeConfig = field(default_factory=BlockSizeConfig)
    value_blocks: BlockSizeConfig = field(default_factory=BlockSizeConfig)
    selection_reasoning: str = ""


@dataclass
class QuantizationConfig:  # :237-260
    precision: QuantizationPrecision = QuantizationPrecision.INT8
    query_precision: QuantizationPrecision = QuantizationPrecision.INT8
    key_precision: QuantizationPrecision = QuantizationPrecision.INT8
    value_precision: QuantizationPrecision = QuantizationPrecision.INT8
    granularity: QuantizationGranularity = QuantizationGranularity.BLOCK_WISE
    block_sizes: BlockSizeConfig = field(default_factory=BlockSizeConfig)
    output_precision: OutputPrecision = OutputPrecision.FP32
    is_causal: bool = False
    scale: Optional[float] = None
    enable_mixed_precision: bool = False
    force_symmetric_quantization: bool = True
    hybrid_strategy: HybridStrategy = HybridStrategy.BALANCED
    enable_per_tensor_granularity: bool = False
    enable_adaptive_block_sizes: bool = False

    def validate_config(self) -> bool:
        ok = self.precision in (QuantizationPrecision.INT8, QuantizationPrecision.INT4)
        return bool(ok and self.force_symmetric_quantization)  # the kernels are symmetric-only (zero point 0)

    def get_recommended_output_precision(self) -> OutputPrecision:
        return OutputPrecision.FP32


# ---- backend registration / dispatch ------------------------------------------------------------------------------
def register_backend() -> None:
    umfa_torch.register_backend()


def unregister_backend() -> None:
    umfa_torch.unregister_backend()


def metal_scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False,
                                       scale: Optional[float] = None, enable_gqa: bool = False):
    return _sdpa.scaled_dot_product_attention(query, key, value, attn_mask, dropout_p, is_causal, scale, enable_gqa)


def rope_scaled_dot_product_attention(query, key, value, rope_cos, rope_sin, attn_mask=None, is_causal: bool = False,
                                      scale: Optional[float] = None):
    return _sdpa.rope_scaled_dot_product_attention(query, key, value, rope_cos, rope_sin, attn_mask, is_causal, scale)


def metal_flash_attention_autograd(query, key, value, is_causal: bool = False, scale: float = 0.0):
    sm = float(scale) if scale else float(query.size(-1)) ** -0.5  # 0.0 = default 1/sqrt(D), .cpp:2704-2707
    return _sdpa._FlashAttentionFn.apply(query, key, value, bool(is_causal), sm)


def metal_quantized_flash_attention_autograd(query, key, value, is_causal: bool = False, scale: float = 0.0,
                                             target_precision: int = 3, quant_mode: int = 0, attn_mask=None):
    sm = float(scale) if scale else float(query.size(-1)) ** -0.5
    prec = QUANT_INT4 if int(target_precision) == 4 else QUANT_INT8
    mode = QUANT_BLOCK_WISE if int(quant_mode) == 2 else QUANT_TENSOR_WISE
    return _sdpa._QuantizedFlashAttentionFn.apply(query, key, value, bool(is_causal), sm, prec, mode, attn_mask)


def set_quantization_mode(precision: int, block_mode: int) -> None:
    umfa_torch.set_quantization_mode(int(precision), int(block_mode))


def clear_quantization_mode() -> None:
    umfa_torch.set_quantization_mode(QUANT_NONE, QUANT_TENSOR_WISE)


get_dispatch_stats = umfa_torch.get_dispatch_stats
reset_dispatch_stats = umfa_torch.reset_dispatch_stats


def hadamard_rotate(tensor: torch.Tensor, block_size: int) -> torch.Tensor:
    return umfa_torch.hadamard_rotate(tensor, int(block_size))


def _quantized(query, key, value, bits: int, blockwise: bool, is_causal: bool, scale, out_dtype):
    if query.dim() != 4:
        raise RuntimeError("quantized SDPA expects [B, H, S, D] tensors")
    o, _ = _ops.quantized_attention_forward(query, key, value, scale=scale, causal=bool(is_causal), bits=bits,
                                            quant_mode="blockwise" if blockwise else "tensorwise")
    return o.to(out_dtype)


def quantized_scaled_dot_product_attention(query, key, value, precision: str = "int8", is_causal: bool = False,
                                           scale: Optional[float] = None):
    bits = 4 if str(precision).lower() == "int4" else 8
    return _quantized(query, key, value, bits, False, is_causal, scale, torch.float32)


def quantized_scaled_dot_product_attention_with_config(query, key, value, config: QuantizationConfig):
    if not config.validate_config():
        raise ValueError("unsupported QuantizationConfig (INT8 / INT4, symmetric only)")
    bits = 4 if config.precision == QuantizationPrecision.INT4 else 8
    blockwise = config.granularity in (QuantizationGranularity.BLOCK_WISE, QuantizationGranularity.HYBRID,
                                       QuantizationGranularity.ROW_WISE)
    return _quantized(query, key, value, bits, blockwise, config.is_causal, config.scale,
                      _OUT_DTYPE[OutputPrecision(config.output_precision)])


# the reference's "enhanced" and "unified" entries end in the same kernel call (MFABridge+Quantized.swift:26-154)
quantized_scaled_dot_product_attention_enhanced = quantized_scaled_dot_product_attention_with_config
quantized_scaled_dot_product_attention_unified = quantized_scaled_dot_product_attention_with_config


# ---- helpers the reference exposes next to the ops --------------------------------------------------------------------
def analyze_tensor_characteristics(tensor: torch.Tensor) -> TensorAnalysisMetrics:
    t = tensor.detach().float()
    amax = float(t.abs().max()) if t.numel() else 0.0
    mean_abs = float(t.abs().mean()) if t.numel() else 0.0
    m = TensorAnalysisMetrics()
    m.dynamic_range = amax / max(float(t.abs()[t != 0].min()) if bool((t != 0).any()) else 1.0, 1e-30)
    m.variance = float(t.var()) if t.numel() > 1 else 0.0
    m.mean_abs_value = mean_abs
    m.tensor_size = t.numel()
    m.memory_footprint = tensor.numel() * tensor.element_size()
    m.sparsity_ratio = float((t == 0).float().mean()) if t.numel() else 0.0
    m.has_outliers = bool(amax > 6.0 * max(float(t.std()), 1e-30)) if t.numel() > 1 else False
    m.quantization_error_estimate = amax / 127.0 / 12 ** 0.5  # rms of a uniform rounding error at the int8 step
    return m


def select_optimal_granularity(tensor: torch.Tensor, precision: QuantizationPrecision = QuantizationPrecision.INT8,
                               strategy: HybridStrategy = HybridStrategy.BALANCED) -> QuantizationGranularity:
    rows = tensor.size(-2) if tensor.dim() >= 2 else 1
    if strategy == HybridStrategy.PERFORMANCE_FIRST and not analyze_tensor_characteristics(tensor).has_outliers:
        return QuantizationGranularity.TENSOR_WISE
    return QuantizationGranularity.BLOCK_WISE if rows >= 64 else QuantizationGranularity.TENSOR_WISE


def determine_output_precision(config: QuantizationConfig, query=None, key=None, value=None) -> OutputPrecision:
    return OutputPrecision(config.output_precision)


def create_typed_output_tensor(reference: torch.Tensor, precision: OutputPrecision, zero_init: bool = False) -> torch.Tensor:
    f = torch.zeros_like if zero_init else torch.empty_like
    return f(reference, dtype=_OUT_DTYPE[OutputPrecision(precision)])


def validate_output_buffer_type(tensor: torch.Tensor, precision: OutputPrecision) -> bool:
    return tensor.dtype == _OUT_DTYPE[OutputPrecision(precision)]


def convert_output_precision(tensor: torch.Tensor, source: OutputPrecision, target: OutputPrecision) -> torch.Tensor:
    return tensor.to(_OUT_DTYPE[OutputPrecision(target)])


def calculate_expected_buffer_size(tensor: torch.Tensor, precision: OutputPrecision) -> int:
    return tensor.numel() * torch.empty((), dtype=_OUT_DTYPE[OutputPrecision(precision)]).element_size()


def is_metal_available() -> bool:  # name kept; on this build it answers "is a supported GPU present"
    return bool(_ffi._lib.mfa_is_device_supported())


def has_native_bfloat() -> bool:
    return bool(_ffi._lib.mfa_has_native_bfloat())


def has_native_bfloat_msl32() -> bool:
    return bool(_ffi._lib.mfa_has_native_bfloat_msl32())


def get_version():
    import ctypes
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _ffi._lib.mfa_get_version(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return (a.value, b.value, c.value)


class MetalSDPABackend:  # python_bindings.cpp:367-373: the static entry points as a class
    register_backend = staticmethod(register_backend)
    unregister_backend = staticmethod(unregister_backend)
    scaled_dot_product_attention = staticmethod(metal_scaled_dot_product_attention)
    rope_scaled_dot_product_attention = staticmethod(rope_scaled_dot_product_attention)
    QUANT_NONE, QUANT_INT8, QUANT_INT4 = QUANT_NONE, QUANT_INT8, QUANT_INT4
    QUANT_TENSOR_WISE, QUANT_BLOCK_WISE = QUANT_TENSOR_WISE, QUANT_BLOCK_WISE


def _not_built(*_a, **_k):
    raise NotImplementedError("MFA Error 3: Device not supported (MLA / sparse indexer are not built on this backend)")


sparse_indexer_scores = _not_built
mla_create_context = mla_destroy_context = mla_init_weights = mla_load_weights = mla_forward = _not_built


class MlaContext:
    def __init__(self, *a, **k):
        _not_built()
