"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares
(no compute calls here).  Error-path behaviour that needs no device is checked too."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "umfa_abi.h"

# the 42 symbols the reference exports + the two MI355X additions
REFERENCE_SYMBOLS = """
mfa_get_quantized_layout mfa_get_quantized_capabilities mfa_create_context mfa_destroy_context
mfa_create_buffer mfa_buffer_from_ptr mfa_buffer_from_ptr_with_strides mfa_buffer_from_mtl_buffer
mfa_buffer_from_mtl_buffer_with_strides mfa_buffer_contents mfa_destroy_buffer mfa_attention_forward
mfa_attention_encode_mtl mfa_attention_forward_quantized mfa_sparse_indexer_scores mfa_attention_backward
mfa_error_string mfa_is_device_supported mfa_get_version mfa_get_gpu_latency
mfa_attention_backward_query_quantized mfa_attention_backward_kv_quantized
mfa_attention_backward_query_quantized_ex mfa_attention_backward_kv_quantized_ex mfa_mla_create_context
mfa_mla_destroy_context mfa_mla_init_weights mfa_mla_load_weights mfa_mla_forward
mfa_attention_forward_str mfa_set_scale_arrays mfa_attention_forward_quantized_unified
mfa_attention_forward_quantized_enhanced mfa_attention_forward_quantized_direct
mfa_multihead_attention_quantized_direct mfa_has_native_bfloat mfa_has_native_bfloat_msl32
mfa_hadamard_rotate mfa_attention_forward_with_lse mfa_quantized_forward_with_lse mfa_quantized_backward
mfa_rope_rotate_encode_mtl
""".split()


@pytest.fixture(scope="module")
def lib():
    import umfa._ffi as ffi
    return ffi._lib


def test_reference_symbol_count():
    assert len(REFERENCE_SYMBOLS) == 42 and len(set(REFERENCE_SYMBOLS)) == 42


def test_header_declares_and_library_exports(lib):
    text = HEADER.read_text()
    declared = set(re.findall(r"\b((?:mfa|umfa)_[a-z0-9_]+)\s*\(", text))
    for name in REFERENCE_SYMBOLS:
        assert name in declared, f"{name} missing from umfa_abi.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared but not exported by libMFAFFI.so"


def test_dynamic_symbol_table_is_the_c_abi_and_nothing_else():
    """`nm -D --defined-only` of libMFAFFI.so = the symbols umfa_abi.h declares: no C++ (umfa::) launchers, no kernel host stubs
    (-fvisibility=hidden + csrc/exports.map)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    so = ROOT / "universal-metal-flash-attention_amd" / "lib" / "libMFAFFI.so"
    out = subprocess.check_output([nm, "-D", "--defined-only", str(so)], text=True)
    names = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    declared = set(re.findall(r"\b((?:mfa|umfa)_[a-z0-9_]+)\s*\(", HEADER.read_text()))
    assert [n for n in names if n not in declared] == []
    assert sum(n.startswith("mfa_") for n in names) == 42 and set(REFERENCE_SYMBOLS) <= set(names)


def test_header_is_plain_c(tmp_path):
    # bindgen/cgo consume the header as C (examples/rust-ffi/build.rs:9-41)
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "mfa_ffi.h"\nint main(void){return sizeof(mfa_quantized_layout_t)==38*4 && '
                   'sizeof(mfa_quantized_capabilities_t)==12 ? 0 : 1;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_version_and_error_strings(lib):
    # MFAFFITests.swift:23-53
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.mfa_get_version(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    assert (a.value, b.value, c.value) == (1, 0, 0)
    import umfa._ffi as ffi
    assert ffi._get_error_string(0) == "Success"
    assert ffi._get_error_string(1) == "Invalid arguments"
    assert ffi._get_error_string(5) == "Execution failed"
    assert ffi._get_error_string(99) == "Unknown error"


def test_quantized_layout_and_capabilities(lib):
    # QuantizedLayoutManifest+FFI.swift:47-49,126-131
    layout = (ctypes.c_int32 * 38)()
    lib.mfa_get_quantized_layout(0, layout)
    assert list(layout) == [-1] * 38

    class Caps(ctypes.Structure):
        _fields_ = [("mh", ctypes.c_bool), ("bw", ctypes.c_bool), ("heads", ctypes.c_uint32), ("blk", ctypes.c_uint32)]

    caps = Caps()
    lib.mfa_get_quantized_capabilities(ctypes.byref(caps))
    assert (caps.mh, caps.bw, caps.heads, caps.blk) == (True, True, 128, 256)


def test_null_handles_are_invalid_args(lib):
    # MFABridge.swift:1105-1110: any required handle NULL -> 1, without touching a device
    rc = lib.mfa_attention_forward(None, None, None, None, None, 1, 4, 4, 1, 4, 0.5, False, 2, 2, 2,
                                   False, False, False, False, None, 0, None, None, 0, 0, 0)
    assert rc == 1
    assert lib.mfa_buffer_contents(None) is None
    lib.mfa_destroy_buffer(None)
    lib.mfa_destroy_context(None)
    assert lib.mfa_get_gpu_latency(None) == 0.0


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import umfa
    assert not umfa.is_device_available()
    with pytest.raises(umfa.MFAError) as e:
        umfa.MFAContext()
    assert e.value.code == 3  # MFA_ERROR_DEVICE_NOT_SUPPORTED: the product path fails loudly


def test_out_of_scope_symbols_return_3(lib):
    lib.mfa_hadamard_rotate.restype = ctypes.c_int32
    assert lib.mfa_hadamard_rotate(None, 64, 1) == 1  # built: NULL buffer -> invalid args
    lib.mfa_sparse_indexer_scores.restype = ctypes.c_int32
    assert lib.mfa_sparse_indexer_scores(None, None, None, 1, 1, 1, 1, 8, ctypes.c_float(1.0), None, None) == 3
    lib.mfa_mla_create_context.restype = ctypes.c_int32
    h = ctypes.c_void_p()
    assert lib.mfa_mla_create_context(ctypes.byref(h)) == 3
