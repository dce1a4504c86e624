"""Helpers mirroring the reference's examples/python-ffi/src/umfa/utils.py."""
from __future__ import annotations

import ctypes
from typing import Tuple

from ._ffi import _lib
from .core import MFAContext


def is_device_available() -> bool:
    """True iff a usable gfx950 (MI355X) device is present (mfa_is_device_supported)."""
    try:
        return bool(_lib.mfa_is_device_supported())
    except Exception:
        return False


# name kept for drop-in callers of the reference package (utils.py:14)
is_metal_available = is_device_available


def get_version() -> Tuple[int, int, int]:
    major, minor, patch = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.mfa_get_version(ctypes.byref(major), ctypes.byref(minor), ctypes.byref(patch))
    return major.value, minor.value, patch.value


def create_context() -> MFAContext:
    if not is_device_available():
        raise RuntimeError("No gfx950 (MI355X) device is available")
    return MFAContext()


def print_system_info() -> None:
    print(f"libMFAFFI version: {'.'.join(map(str, get_version()))}")
    print(f"gfx950 device available: {is_device_available()}")
    print(f"native bfloat16: {bool(_lib.mfa_has_native_bfloat())}")
