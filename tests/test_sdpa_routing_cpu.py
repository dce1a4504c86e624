"""Host-side routing logic of umfa_torch.scaled_dot_product_attention that needs no GPU
(metal_sdpa_backend.cpp:1643-1904 semantics: promotion, GQA, fallback conditions, counters)."""
import pytest
import torch
import torch.nn.functional as F

import umfa_torch
from umfa_torch import sdpa


@pytest.fixture(autouse=True)
def _reset():
    umfa_torch.reset_dispatch_stats()
    umfa_torch.set_quantization_mode(0, 0)
    yield
    umfa_torch.unregister_backend()
    umfa_torch.set_quantization_mode(0, 0)


def test_cpu_tensors_fall_back_to_native_and_count():
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 2, 16, 8) for _ in range(3))
    out = umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert torch.allclose(out, sdpa._native_sdpa(q, k, v, is_causal=True))
    s = umfa_torch.get_dispatch_stats()
    assert s["total"] == 1 and s["pytorch_fallback"] == 1 and s["fp32_instream"] == 0


def test_2d_and_3d_promotion_squeeze_back():
    q, k, v = (torch.randn(16, 8) for _ in range(3))
    out = umfa_torch.scaled_dot_product_attention(q, k, v, scale=0.5)
    assert out.shape == (16, 8)
    assert torch.allclose(out, sdpa._native_sdpa(q, k, v, scale=0.5), atol=1e-6)
    q3, k3, v3 = (torch.randn(3, 16, 8) for _ in range(3))
    assert umfa_torch.scaled_dot_product_attention(q3, k3, v3).shape == (3, 16, 8)
    assert umfa_torch.get_dispatch_stats()["total"] == 2  # counted once per call, after promotion


def test_fp16_mask_is_promoted_for_the_native_fallback():
    q, k, v = (torch.randn(1, 1, 8, 8) for _ in range(3))
    m = torch.zeros(8, 8, dtype=torch.float16)
    out = umfa_torch.scaled_dot_product_attention(q, k, v, attn_mask=m)  # native CPU SDPA needs a float mask
    assert torch.allclose(out, sdpa._native_sdpa(q, k, v), atol=1e-6)


def test_kv_length_mismatch_raises():
    with pytest.raises(RuntimeError):
        umfa_torch.scaled_dot_product_attention(torch.randn(1, 1, 4, 4), torch.randn(1, 1, 5, 4), torch.randn(1, 1, 6, 4))


def test_register_and_context_manager():
    native = F.scaled_dot_product_attention
    with umfa_torch.use_umfa_sdpa():
        assert F.scaled_dot_product_attention is umfa_torch.scaled_dot_product_attention
    assert F.scaled_dot_product_attention is native
    umfa_torch.register_backend()
    assert F.scaled_dot_product_attention is umfa_torch.scaled_dot_product_attention
    umfa_torch.unregister_backend()
    assert F.scaled_dot_product_attention is native


def test_quantization_mode_validation():
    umfa_torch.set_quantization_mode(umfa_torch.QUANT_INT8, umfa_torch.QUANT_BLOCK_WISE)
    assert umfa_torch.get_quantization_mode() == (3, 2)
    with pytest.raises(ValueError):
        umfa_torch.set_quantization_mode(7, 0)
    with pytest.raises(ValueError):
        umfa_torch.set_quantization_mode(3, 1)


def test_counter_names_match_reference():
    # metal_sdpa_backend.h:666-681
    assert set(umfa_torch.get_dispatch_stats()) == {
        "total", "quantized_autograd", "fp32_autograd", "fp32_direct", "fp32_instream", "rope_instream",
        "rope_autograd", "pytorch_fallback", "mask_all_true_skipped"}


def test_custom_op_traces_as_one_node_with_fake_tensors():
    """umfa::sdpa_forward has a fake implementation: a trace over fake ROCm tensors (what torch.compile does) keeps it
    as one opaque node with the right output metadata -- no GPU needed to check that."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.fx.experimental.proxy_tensor import make_fx

    import umfa_torch
    with FakeTensorMode() as mode:
        q = torch.empty(2, 4, 256, 128, device="cuda", dtype=torch.bfloat16)
        gm = make_fx(lambda a, b, c: umfa_torch.library.sdpa(a, b, c, is_causal=True), tracing_mode="fake")(q, q, q)
    targets = [n.target for n in gm.graph.nodes if n.op == "call_function"]
    assert torch.ops.umfa.sdpa_forward.default in targets, targets
    assert not any("scaled_dot_product" in str(t) for t in targets)
    # dropout is not ours: the same entry point traces to torch's own kernels instead
    with FakeTensorMode():
        q = torch.empty(2, 4, 256, 128, device="cuda", dtype=torch.bfloat16)
        assert not umfa_torch.library.op_supports(q, q, q, None, 0.1, False)
        assert not umfa_torch.library.op_supports(q, q, q, torch.empty(3, 256, dtype=torch.bool, device="cuda"), 0.0, False)
