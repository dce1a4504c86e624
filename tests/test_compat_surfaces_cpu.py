"""The Python surfaces an existing user of the reference imports by name: `metal_sdpa_extension`
(the pybind11 module, python_bindings.cpp:40-426) and `pytorch_custom_op_ffi` (__init__.py:21-39).  CPU-only: names,
defaults and enum values; the compute behind them is covered by tests/test_gpu_compat_surfaces.py."""
import inspect

import pytest

torch = pytest.importorskip("torch")

PYBIND_NAMES = [  # every m.def / m.attr / class of python_bindings.cpp, in file order
    "register_backend", "unregister_backend", "metal_scaled_dot_product_attention", "rope_scaled_dot_product_attention",
    "metal_flash_attention_autograd", "metal_quantized_flash_attention_autograd", "set_quantization_mode",
    "clear_quantization_mode", "get_dispatch_stats", "reset_dispatch_stats", "QUANT_NONE", "QUANT_INT8", "QUANT_INT4",
    "QUANT_TENSOR_WISE", "QUANT_BLOCK_WISE", "hadamard_rotate", "quantized_scaled_dot_product_attention",
    "quantized_scaled_dot_product_attention_with_config", "quantized_scaled_dot_product_attention_enhanced",
    "quantized_scaled_dot_product_attention_unified", "sparse_indexer_scores", "QuantizationPrecision",
    "QuantizationGranularity", "HybridStrategy", "OutputPrecision", "BlockSizeConfig", "TensorAnalysisMetrics",
    "HybridGranularityConfig", "QuantizationConfig", "analyze_tensor_characteristics", "select_optimal_granularity",
    "determine_output_precision", "create_typed_output_tensor", "validate_output_buffer_type", "convert_output_precision",
    "calculate_expected_buffer_size", "is_metal_available", "has_native_bfloat", "has_native_bfloat_msl32", "get_version",
    "MetalSDPABackend", "MlaContext", "mla_create_context", "mla_destroy_context", "mla_init_weights", "mla_load_weights",
    "mla_forward"]


def test_extension_module_exports_every_reference_name():
    import metal_sdpa_extension as ext
    missing = [n for n in PYBIND_NAMES if not hasattr(ext, n)]
    assert not missing, missing


def test_signatures_match_the_pybind_defaults():
    import metal_sdpa_extension as ext
    sig = inspect.signature(ext.metal_scaled_dot_product_attention)
    assert list(sig.parameters) == ["query", "key", "value", "attn_mask", "dropout_p", "is_causal", "scale", "enable_gqa"]
    assert sig.parameters["dropout_p"].default == 0.0 and sig.parameters["is_causal"].default is False
    sig = inspect.signature(ext.rope_scaled_dot_product_attention)
    assert list(sig.parameters) == ["query", "key", "value", "rope_cos", "rope_sin", "attn_mask", "is_causal", "scale"]
    sig = inspect.signature(ext.metal_quantized_flash_attention_autograd)
    assert sig.parameters["target_precision"].default == 3 and sig.parameters["quant_mode"].default == 0
    sig = inspect.signature(ext.quantized_scaled_dot_product_attention)
    assert sig.parameters["precision"].default == "int8"
    sig = inspect.signature(ext.metal_flash_attention_autograd)
    assert sig.parameters["scale"].default == 0.0


def test_enums_and_constants():
    import metal_sdpa_extension as ext
    assert [int(ext.QuantizationPrecision[n]) for n in ("FP16", "BF16", "FP32", "INT8", "INT4")] == [0, 1, 2, 3, 4]
    assert set(ext.QuantizationGranularity.__members__) == {"TENSOR_WISE", "ROW_WISE", "BLOCK_WISE", "HYBRID"}
    assert set(ext.HybridStrategy.__members__) == {"PERFORMANCE_FIRST", "ACCURACY_FIRST", "BALANCED"}
    assert set(ext.OutputPrecision.__members__) == {"FP32", "FP16", "BF16"}
    assert (ext.QUANT_NONE, ext.QUANT_INT8, ext.QUANT_INT4) == (0, 3, 4) or ext.QUANT_INT8 != ext.QUANT_INT4
    cfg = ext.QuantizationConfig()
    assert cfg.validate_config() and cfg.get_recommended_output_precision() == ext.OutputPrecision.FP32
    cfg.precision = ext.QuantizationPrecision.FP16
    assert not cfg.validate_config()
    b = ext.BlockSizeConfig(32, 64, 64, 0)
    assert (b.query_block_size, b.key_block_size) == (32, 64)


def test_out_of_scope_operators_say_so():
    import metal_sdpa_extension as ext
    for fn in (ext.sparse_indexer_scores, ext.mla_create_context, ext.mla_forward):
        with pytest.raises(NotImplementedError):
            fn()
    with pytest.raises(NotImplementedError):
        ext.MlaContext()


def test_tensor_helpers_on_cpu():
    import metal_sdpa_extension as ext
    t = torch.randn(4, 8)
    m = ext.analyze_tensor_characteristics(t)
    assert m.tensor_size == 32 and m.memory_footprint == 128 and m.mean_abs_value > 0
    out = ext.create_typed_output_tensor(t, ext.OutputPrecision.BF16, True)
    assert out.dtype == torch.bfloat16 and ext.validate_output_buffer_type(out, ext.OutputPrecision.BF16)
    assert ext.calculate_expected_buffer_size(t, ext.OutputPrecision.FP16) == 64
    assert ext.convert_output_precision(t, ext.OutputPrecision.FP32, ext.OutputPrecision.FP16).dtype == torch.float16
    assert ext.select_optimal_granularity(torch.randn(128, 64)) == ext.QuantizationGranularity.BLOCK_WISE


def test_package_surface():
    import pytorch_custom_op_ffi as pkg
    for n in ("register_metal_sdpa_backend", "unregister_metal_sdpa_backend", "use_metal_sdpa", "is_metal_sdpa_available",
              "metal_sdpa_version", "MetalSDPAContext"):
        assert hasattr(pkg, n), n
    assert hasattr(torch.backends, "metal_sdpa")
    assert torch.backends.metal_sdpa.enabled is False
    assert pkg.metal_sdpa_version() == (1, 0, 0)  # metal_sdpa_backend.cpp:1601-1610
    if not torch.cuda.is_available():
        assert pkg.is_metal_sdpa_available() is False
        with pytest.raises(RuntimeError):
            pkg.register_metal_sdpa_backend()
