"""pytorch_custom_op_ffi -- ROCm drop-in for the reference's Python package of the same name
(examples/pytorch-custom-op-ffi/python/pytorch_custom_op_ffi/__init__.py:21-39): same public names."""
from .backend import (MetalSDPAContext, is_metal_sdpa_available, metal_sdpa_version, register_metal_sdpa_backend,
                      unregister_metal_sdpa_backend, use_metal_sdpa)

__version__ = "0.1.0"
__all__ = ["register_metal_sdpa_backend", "unregister_metal_sdpa_backend", "use_metal_sdpa", "is_metal_sdpa_available",
           "metal_sdpa_version", "MetalSDPAContext"]
