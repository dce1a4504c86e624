"""umfa -- numpy/ctypes binding of libMFAFFI.so for AMD Instinct MI355X.

Same public surface as the reference's `umfa` package
(examples/python-ffi/src/umfa/__init__.py): MFAContext, MFABuffer,
flash_attention_forward, attention, quantized_attention, MFAError, create_context,
is_metal_available (alias of is_device_available), get_version, print_system_info.
"""
from ._ffi import MFAError
from .core import (MFABuffer, MFAContext, attention, attention_backward, flash_attention_forward,
                   quantized_attention)
from .sdpa_replacement import MetalSDPA
from .utils import create_context, get_version, is_device_available, is_metal_available, print_system_info

__version__ = "1.0.0"

__all__ = ["MFAContext", "MFABuffer", "flash_attention_forward", "attention", "attention_backward",
           "quantized_attention", "MetalSDPA", "MFAError", "create_context", "is_metal_available", "is_device_available",
           "get_version", "print_system_info", "__version__"]
