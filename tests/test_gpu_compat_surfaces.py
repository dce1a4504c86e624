"""The reference's Python entry points (metal_sdpa_extension / pytorch_custom_op_ffi names) running on the MI355X."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402


def ref64(q, k, v, **kw):
    return F.scaled_dot_product_attention(q.double().cpu(), k.double().cpu(), v.double().cpu(), **kw)


def test_extension_entry_points():
    import metal_sdpa_extension as ext
    assert ext.is_metal_available() and ext.get_version() == (1, 0, 0)
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, 4, 256, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    ext.reset_dispatch_stats()
    o = ext.metal_scaled_dot_product_attention(q, k, v, is_causal=True)
    assert o.dtype == torch.bfloat16 and (o.float().cpu() - ref64(q, k, v, is_causal=True)).abs().max() < 3e-2
    assert ext.get_dispatch_stats()["fp32_instream"] == 1
    # autograd entry (scale 0.0 = default)
    qg, kg, vg = (t.float().requires_grad_(True) for t in (q, k, v))
    ext.metal_flash_attention_autograd(qg, kg, vg).sum().backward()
    assert qg.grad is not None and torch.isfinite(qg.grad).all()
    # quantised entries: string precision and config object
    o8 = ext.quantized_scaled_dot_product_attention(q, k, v, precision="int8")
    assert o8.dtype == torch.float32 and (o8.cpu() - ref64(q, k, v)).abs().max() < 5e-2
    cfg = ext.QuantizationConfig()
    cfg.output_precision = ext.OutputPrecision.BF16
    cfg.is_causal = True
    o8c = ext.quantized_scaled_dot_product_attention_unified(q, k, v, cfg)
    assert o8c.dtype == torch.bfloat16 and (o8c.float().cpu() - ref64(q, k, v, is_causal=True)).abs().max() < 6e-2
    # quantisation mode routes F.sdpa-style calls through the quantised autograd path
    ext.set_quantization_mode(ext.QUANT_INT8, ext.QUANT_BLOCK_WISE)
    try:
        oq = ext.metal_scaled_dot_product_attention(q, k, v)
        assert ext.get_dispatch_stats()["quantized_autograd"] == 1
        assert (oq.float().cpu() - ref64(q, k, v)).abs().max() < 6e-2
    finally:
        ext.clear_quantization_mode()
    # Hadamard: orthonormal, its own inverse
    x = torch.randn(64, 128, device="cuda")
    y = ext.hadamard_rotate(x.clone(), 128)
    assert (ext.hadamard_rotate(y.clone(), 128) - x).abs().max() < 1e-4
    assert abs(float(y.norm()) - float(x.norm())) < 1e-2


def test_package_context_managers():
    import pytorch_custom_op_ffi as pkg
    assert pkg.is_metal_sdpa_available()
    torch.manual_seed(1)
    q, k, v = (torch.randn(1, 2, 128, 64, device="cuda", dtype=torch.float16) for _ in range(3))
    native = F.scaled_dot_product_attention
    with pkg.use_metal_sdpa() as dev:
        assert dev.type == "cuda" and torch.backends.metal_sdpa.enabled
        o = F.scaled_dot_product_attention(q, k, v)
    assert F.scaled_dot_product_attention is native and not torch.backends.metal_sdpa.enabled
    assert (o.float().cpu() - ref64(q, k, v)).abs().max() < 5e-3
    with pkg.MetalSDPAContext() as ctx:
        out = ctx.direct_call(q.cpu(), k.cpu(), v.cpu(), is_causal=True)  # CPU tensors in, CPU tensor out
        assert out.device.type == "cpu" and out.dtype == torch.float16
        assert (out.double() - ref64(q, k, v, is_causal=True)).abs().max() < 5e-3
    pkg.unregister_metal_sdpa_backend()
    torch.backends.metal_sdpa.enabled = True
    assert F.scaled_dot_product_attention is not native
    torch.backends.metal_sdpa.enabled = False
    assert F.scaled_dot_product_attention is native
