"""ROCm counterpart of the reference package's backend module (its public names: examples/pytorch-custom-op-ffi/python/
pytorch_custom_op_ffi/backend.py:35-330), written around ONE state object instead of the reference's module-level flags.

`_Switch` owns the "is our SDPA installed in torch" state; every public name of the reference is a thin view of it:
the two registration functions, the `use_metal_sdpa` context manager, `MetalSDPAContext` and the
`torch.backends.metal_sdpa` config object.  The execution device is `cuda` (= the MI355X) where the reference says `mps`.
One deliberate behavioural difference: `MetalSDPAContext.direct_call` runs fp16 / bf16 inputs on the HIP kernels (the
reference computes them with torch's own SDPA in fp32 without saying so, backend.py:202-213)."""
from __future__ import annotations

import threading
from typing import Optional, Tuple

import torch

import metal_sdpa_extension as _ext


class _Switch:
    """Installed / not installed, guarded by one lock; `hold()` is a scoped install that leaves a pre-existing one alone."""

    def __init__(self) -> None:
        self._lock = threading.Lock()
        self.on = False

    @staticmethod
    def usable() -> bool:
        try:
            return bool(torch.cuda.is_available() and _ext.is_metal_available())
        except Exception:  # noqa: BLE001  a broken install answers "not available", as the reference does
            return False

    @staticmethod
    def device() -> torch.device:
        if not torch.cuda.is_available():
            raise RuntimeError("Metal SDPA backend (MI355X build) requires a GPU")
        return torch.device("cuda", torch.cuda.current_device())

    def set(self, want: bool) -> None:
        with self._lock:
            if want == self.on:
                return
            if want:
                if not self.usable():
                    raise RuntimeError("Metal SDPA backend (MI355X build) needs a supported GPU")
                _ext.register_backend()
            else:
                _ext.unregister_backend()
            self.on = want

    class _Hold:
        def __init__(self, switch: "_Switch") -> None:
            self._switch, self._mine = switch, False

        def __enter__(self) -> torch.device:
            self._mine = not self._switch.on
            if self._mine:
                self._switch.set(True)
            return self._switch.device()

        def __exit__(self, *exc) -> None:
            if self._mine:
                self._switch.set(False)

    def hold(self) -> "_Switch._Hold":
        return _Switch._Hold(self)


_switch = _Switch()


# ---- the reference's function names ---------------------------------------------------------------------------------
def is_metal_sdpa_available() -> bool:
    return _switch.usable()


def metal_sdpa_version() -> Optional[Tuple[int, int, int]]:
    try:
        return _ext.get_version()
    except Exception:  # noqa: BLE001
        return None


def register_metal_sdpa_backend() -> None:
    _switch.set(True)


def unregister_metal_sdpa_backend() -> None:
    _switch.set(False)


def use_metal_sdpa():
    """`with use_metal_sdpa() as device:` -- our SDPA for the block (left installed if it already was), yields the device."""
    return _switch.hold()


class MetalSDPAContext:
    """Scope with a current execution device and explicit tensor movement helpers; `direct_call` bypasses the torch patch."""

    def __init__(self, auto_register: bool = True):
        self.auto_register = auto_register
        self.device: Optional[torch.device] = None

    def __enter__(self) -> "MetalSDPAContext":
        if self.auto_register:
            _switch.set(True)
        self.device = _switch.device()
        return self

    def __exit__(self, *exc) -> None:
        self.device = None

    def _active(self) -> torch.device:
        if self.device is None:
            raise RuntimeError("MetalSDPAContext is not active")
        return self.device

    def to_device(self, tensor: torch.Tensor) -> torch.Tensor:
        return tensor.to(self._active())

    @staticmethod
    def to_cpu(tensor: torch.Tensor) -> torch.Tensor:
        return tensor.cpu()

    def direct_call(self, query, key, value, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False,
                    scale: Optional[float] = None) -> torch.Tensor:
        dev = self._active()
        home, dtype = query.device, query.dtype
        on_dev = [None if t is None else t.to(dev) for t in (query, key, value, attn_mask)]
        out = _ext.metal_scaled_dot_product_attention(on_dev[0], on_dev[1], on_dev[2], on_dev[3], dropout_p, is_causal, scale)
        return out.to(device=home, dtype=dtype)


class MetalSDPABackendConfig:
    """`torch.backends.metal_sdpa`: `.enabled` (read / write), `.available`, `.version`."""

    enabled = property(lambda self: _switch.on, lambda self, value: _switch.set(bool(value)))
    available = property(lambda self: _switch.usable())
    version = property(lambda self: metal_sdpa_version())


if not hasattr(torch.backends, "metal_sdpa"):
    torch.backends.metal_sdpa = MetalSDPABackendConfig()
