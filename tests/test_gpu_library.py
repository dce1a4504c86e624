"""Dispatcher-level binding (umfa_torch/library.py): the umfa::sdpa_forward / umfa::sdpa_backward custom ops under
torch.compile(fullgraph=True), and the opt-in override of aten::scaled_dot_product_attention for the CUDA keys
(reference: TORCH_LIBRARY_IMPL(aten, MPS, m), metal_sdpa_backend.cpp:3464-3470)."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


def ref64(q, k, v, **kw):
    return F.scaled_dot_product_attention(q.double().cpu(), k.double().cpu(), v.double().cpu(), **kw)


class Block(torch.nn.Module):
    def __init__(self, causal):
        super().__init__()
        self.causal = causal

    def forward(self, q, k, v):
        return F.scaled_dot_product_attention(q * 1.0, k, v, is_causal=self.causal) + 0.0


@pytest.fixture()
def backend():
    import umfa_torch
    umfa_torch.register_backend()
    umfa_torch.reset_dispatch_stats()
    yield umfa_torch
    umfa_torch.unregister_backend()
    umfa_torch.library.override_aten_sdpa(False)


def test_compile_fullgraph_dispatches_to_the_hip_kernels(backend):
    q, k, v = (torch.randn(2, 4, 512, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    mod = torch.compile(Block(True), fullgraph=True)
    out = mod(q, k, v)
    torch.cuda.synchronize()
    assert backend.last_kernel().startswith("fa_fwd16"), backend.last_kernel()
    st = backend.get_dispatch_stats()
    assert st["total"] == 1 and st["fp32_instream"] == 1 and st["pytorch_fallback"] == 0
    assert out.dtype == torch.bfloat16 and (out.double().cpu() - ref64(q, k, v, is_causal=True)).abs().max() < 2e-2
    before = backend.get_dispatch_stats()["fp32_instream"]  # (ref64 went through the patched F.sdpa too: CPU -> fallback)
    out2 = mod(q, k, v)  # second call of the compiled graph: one more executed attention
    assert torch.equal(out, out2) and backend.get_dispatch_stats()["fp32_instream"] == before + 1


def test_compile_fullgraph_training_step(backend):
    q, k, v = (torch.randn(1, 4, 256, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    mod = torch.compile(Block(False), fullgraph=True)
    out = mod(q, k, v)
    out.float().square().sum().backward()
    assert backend.last_kernel().startswith("fa_bwd"), backend.last_kernel()
    qd, kd, vd = (t.detach().double().cpu().requires_grad_(True) for t in (q, k, v))
    F.scaled_dot_product_attention(qd, kd, vd).square().sum().backward()
    for g, r in ((q.grad, qd.grad), (k.grad, kd.grad), (v.grad, vd.grad)):
        assert float((g.double().cpu() - r).abs().max() / r.abs().max()) < 3e-2


def test_custom_op_opcheck_and_direct_call():
    import umfa_torch  # noqa: F401
    q, k, v = (torch.randn(1, 2, 128, 64, device="cuda", dtype=torch.float16) for _ in range(3))
    torch.library.opcheck(torch.ops.umfa.sdpa_forward, (q, k, v, None, True, 0.125),
                          test_utils=("test_schema", "test_faketensor"))
    mask = torch.ones(1, 1, 128, 128, dtype=torch.bool, device="cuda").tril(5)
    out = umfa_torch.library.sdpa(q, k, v, attn_mask=mask)
    assert (out.double().cpu() - ref64(q, k, v, attn_mask=mask.cpu())).abs().max() < 2e-3


def test_aten_override_reaches_callers_that_captured_sdpa_early(backend):
    backend.unregister_backend()
    captured = F.scaled_dot_product_attention  # a caller that bound the function before we registered anything
    assert captured is torch._C._nn.scaled_dot_product_attention or captured.__module__ != "umfa_torch.sdpa"
    q, k, v = (torch.randn(1, 4, 384, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    backend.library.override_aten_sdpa(True)
    backend.reset_dispatch_stats()
    out = captured(q, k, v, is_causal=True)
    torch.cuda.synchronize()
    st = backend.get_dispatch_stats()
    assert st["total"] == 1 and st["fp32_instream"] == 1 and backend.last_kernel().startswith("fa_fwd16")
    assert (out.double().cpu() - ref64(q, k, v, is_causal=True)).abs().max() < 2e-2
    # unsupported call (dropout): falls through to torch's composite kernel without re-entering the override
    out_d = captured(q, k, v, dropout_p=0.5)
    assert out_d.shape == q.shape and backend.get_dispatch_stats()["pytorch_fallback"] == 1
    # training through the override (AutogradCUDA key): our backward kernels
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
    captured(qg, kg, vg).float().sum().backward()
    assert backend.last_kernel().startswith("fa_bwd") and qg.grad is not None and torch.isfinite(qg.grad).all()
    # CPU tensors are not ours
    c = torch.randn(1, 2, 16, 8)
    assert captured(c, c, c).shape == c.shape
    backend.library.override_aten_sdpa(False)
    backend.reset_dispatch_stats()
    captured(q, k, v)
    assert backend.get_dispatch_stats()["total"] == 0
