"""umfa_torch -- PyTorch-ROCm binding of the MI355X flash-attention kernels (in-stream, zero-copy)."""
from .ops import (attention_encode, attention_forward, bench_int8, context, gpu_latency, hadamard_rotate, last_kernel, options, release_scratch, set_option, get_option, pv_fp16_status,
                  quantized_attention_forward, quantized_attention_forward_stream, quantized_attention_backward_stream, attention_backward, rope_rotate,
                  rope_attention_forward)
from . import library, parallel
from .sdpa import (QUANT_BLOCK_WISE, QUANT_INT4, QUANT_INT8, QUANT_NONE, QUANT_TENSOR_WISE, get_dispatch_stats,
                   get_quantization_mode, register_backend, reset_dispatch_stats, rope_scaled_dot_product_attention,
                   scaled_dot_product_attention,
                   set_quantization_mode, unregister_backend, use_umfa_sdpa)

__all__ = ["library", "parallel", "rope_rotate", "rope_attention_forward", "hadamard_rotate", "attention_forward", "attention_encode", "quantized_attention_forward", "quantized_attention_forward_stream", "quantized_attention_backward_stream", "attention_backward", "bench_int8", "context",
           "gpu_latency", "last_kernel", "set_option", "get_option", "pv_fp16_status", "options", "release_scratch", "scaled_dot_product_attention", "rope_scaled_dot_product_attention", "register_backend", "unregister_backend",
           "use_umfa_sdpa", "set_quantization_mode", "get_quantization_mode", "get_dispatch_stats",
           "reset_dispatch_stats", "QUANT_NONE", "QUANT_INT8", "QUANT_INT4", "QUANT_TENSOR_WISE", "QUANT_BLOCK_WISE"]
