"""Mirror of the reference's pytorch_custom_op_ffi/backend.py (:35-330) for PyTorch-ROCm: registration helpers, the
`use_metal_sdpa` context manager, `MetalSDPAContext` and the `torch.backends.metal_sdpa` config object.  The
execution device is `cuda` (= the MI355X) instead of `mps`.  One deliberate difference: `direct_call` runs fp16 / bf16
inputs through the HIP kernels (the reference silently computes them with torch SDPA in fp32, backend.py:202-213)."""
from __future__ import annotations

import threading
from contextlib import contextmanager
from typing import Optional, Tuple

import torch

import metal_sdpa_extension as _ext

_backend_registered = False
_registration_lock = threading.Lock()


def is_metal_sdpa_available() -> bool:
    try:
        return torch.cuda.is_available() and _ext.is_metal_available()
    except Exception:
        return False


def metal_sdpa_version() -> Optional[Tuple[int, int, int]]:
    try:
        return _ext.get_version()
    except Exception:
        return None


def register_metal_sdpa_backend() -> None:
    global _backend_registered
    with _registration_lock:
        if _backend_registered:
            return
        if not is_metal_sdpa_available():
            raise RuntimeError("Metal SDPA backend (MI355X build) needs a supported GPU")
        _ext.register_backend()
        _backend_registered = True


def unregister_metal_sdpa_backend() -> None:
    global _backend_registered
    with _registration_lock:
        if not _backend_registered:
            return
        _ext.unregister_backend()
        _backend_registered = False


def _resolve_execution_device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    raise RuntimeError("Metal SDPA backend (MI355X build) requires a GPU")


@contextmanager
def use_metal_sdpa():
    was = _backend_registered
    if not was:
        register_metal_sdpa_backend()
    try:
        yield _resolve_execution_device()
    finally:
        if not was:
            unregister_metal_sdpa_backend()


class MetalSDPAContext:
    def __init__(self, auto_register: bool = True):
        self.auto_register = auto_register
        self.device = None

    def __enter__(self):
        if self.auto_register and not _backend_registered:
            register_metal_sdpa_backend()
        self.device = _resolve_execution_device()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.device = None

    def to_device(self, tensor: torch.Tensor) -> torch.Tensor:
        if self.device is None:
            raise RuntimeError("Context not active")
        return tensor.to(self.device)

    def to_cpu(self, tensor: torch.Tensor) -> torch.Tensor:
        return tensor.cpu()

    def direct_call(self, query, key, value, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False,
                    scale: Optional[float] = None) -> torch.Tensor:
        if self.device is None:
            raise RuntimeError("MetalSDPAContext is not active")
        orig_device, orig_dtype = query.device, query.dtype
        q, k, v = (t.to(self.device) for t in (query, key, value))
        m = attn_mask.to(self.device) if attn_mask is not None else None
        out = _ext.metal_scaled_dot_product_attention(q, k, v, m, dropout_p, is_causal, scale)
        return out.to(device=orig_device, dtype=orig_dtype)


class MetalSDPABackendConfig:
    @property
    def enabled(self) -> bool:
        return _backend_registered

    @enabled.setter
    def enabled(self, value: bool):
        if value and not _backend_registered:
            register_metal_sdpa_backend()
        elif not value and _backend_registered:
            unregister_metal_sdpa_backend()

    @property
    def available(self) -> bool:
        return is_metal_sdpa_available()

    @property
    def version(self):
        return metal_sdpa_version()


if not hasattr(torch.backends, "metal_sdpa"):
    torch.backends.metal_sdpa = MetalSDPABackendConfig()
