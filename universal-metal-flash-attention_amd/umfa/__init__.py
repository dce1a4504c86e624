"""umfa -- numpy/ctypes binding of libMFAFFI.so for AMD Instinct MI355X.

The package answers to the names the reference's `umfa` package exports (examples/python-ffi/src/umfa/__init__.py), plus
`attention_backward`, `MetalSDPA` (the reference's examples/pytorch_sdpa_replacement.py wrapper) and `is_device_available`.
"""
from . import _ffi as _f, core as _c, sdpa_replacement as _s, utils as _u

__version__ = "1.0.0"

_EXPORTS = {
    _c: ("MFAContext", "MFABuffer", "flash_attention_forward", "attention", "attention_backward", "quantized_attention"),
    _s: ("MetalSDPA",),
    _f: ("MFAError",),
    _u: ("create_context", "is_metal_available", "is_device_available", "get_version", "print_system_info"),
}
for _mod, _names in _EXPORTS.items():
    for _n in _names:
        globals()[_n] = getattr(_mod, _n)
__all__ = [n for names in _EXPORTS.values() for n in names] + ["__version__"]
del _mod, _names, _n
