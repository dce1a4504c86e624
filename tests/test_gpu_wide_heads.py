"""Head dims 257 ... 1024 (fa_fwd_wide.hip): the reference's callers admit head_dim <= 1024
(examples/pytorch-custom-op-ffi/src/metal_sdpa_backend.cpp:1078-1086, :1382-1384); until round 5 every entry here refused them.
fp32 arithmetic for every operand type, so the bar is the fp32-exact kernel's: 1e-5 max-abs on fp32 inputs (the reference's own fp32
tolerance, tests/test_scale_factor_fix.py:66), the operand format's rounding on 16-bit ones -- through the blocking C ABI, the in-stream
entry and the torch SDPA surface, with masks, causal, ragged sizes, strides, LSE."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _oracle():
    from oracle import oracle
    return oracle


def npy(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.detach().cpu().contiguous().numpy()


@pytest.mark.parametrize("D", [320, 264, 512, 640, 1024])
@pytest.mark.parametrize("causal", [False, True])
def test_fp32_host_arrays_through_the_blocking_abi(D, causal):
    import umfa
    rng = np.random.default_rng(D)
    B, H, Sq, Skv = 1, 2, 70, 101
    q = rng.standard_normal((B, H, Sq, D)).astype(np.float32)
    k = rng.standard_normal((B, H, Skv, D)).astype(np.float32)
    v = rng.standard_normal((B, H, Skv, D)).astype(np.float32)
    with umfa.MFAContext() as ctx:
        o = umfa.flash_attention_forward(ctx, q, k, v, causal=causal, input_precision="fp32", intermediate_precision="fp32", layout="bhsd")
        assert ctx.last_kernel == ("fa_fwd_wide<512>" if D <= 512 else "fa_fwd_wide<1024>"), ctx.last_kernel
    ref = _oracle().sdpa_forward(q, k, v, causal=causal)
    assert float(np.abs(o - ref).max()) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [320, 1024])
def test_in_stream_entry_masks_strides_lse(dtype, D):
    import umfa_torch
    torch.manual_seed(D)
    B, H, Sq, Skv = 2, 3, 97, 160
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dtype)
    k = torch.randn(B, Skv, H, D, device="cuda", dtype=dtype).transpose(1, 2)  # strided K
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dtype)
    orc = _oracle()
    tol = 1e-5 if dtype == torch.float32 else 2e-5  # fp32 arithmetic on the rounded operands: nothing is rounded in between
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    assert umfa_torch.last_kernel().startswith("fa_fwd_wide<")
    ref, ref_lse = orc.sdpa_forward(npy(q), npy(k), npy(v), return_lse=True)
    assert float(np.abs(o.cpu().numpy() - ref).max()) < tol
    assert float(np.abs(lse.cpu().numpy().reshape(ref_lse.shape) - ref_lse).max()) < 1e-4
    mask = torch.rand(1, H, Sq, Skv, device="cuda") > 0.4
    mask[0, 1, 5] = False  # a row that attends to nothing: O = 0
    om = umfa_torch.attention_forward(q, k, v, mask=mask, out_dtype=torch.float32)
    refm = orc.sdpa_forward(npy(q), npy(k), npy(v), mask=mask.cpu().numpy(), mask_type=orc.MASK_BOOL)
    assert float(np.abs(om.cpu().numpy() - refm).max()) < tol and float(om[:, 1, 5].abs().max()) == 0.0
    bias = (torch.randn(Sq, Skv, device="cuda") * 2).to(torch.float32)
    ob = umfa_torch.attention_forward(q, k, v, mask=bias, causal=True, out_dtype=torch.float32)
    refb = orc.sdpa_forward(npy(q), npy(k), npy(v), causal=True, mask=bias.cpu().numpy(), mask_type=orc.MASK_ADDITIVE)
    assert float(np.abs(ob.cpu().numpy() - refb).max()) < tol
    ow = umfa_torch.attention_forward(q, k, v, window=(20, 7), out_dtype=torch.float32)
    r, c = torch.arange(Sq)[:, None], torch.arange(Skv)[None, :]
    refw = orc.sdpa_forward(npy(q), npy(k), npy(v), mask=((c >= r - 20) & (c <= r + 7)).numpy(), mask_type=orc.MASK_BOOL)
    assert float(np.abs(ow.cpu().numpy() - refw).max()) < tol
    if dtype != torch.float32:  # output in the operand type (the torch caller's cast-back, fused)
        o16 = umfa_torch.attention_forward(q, k, v)
        assert o16.dtype == dtype and float((o16.float() - o).abs().max()) <= float(o.abs().max()) * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)


def test_torch_sdpa_surface_takes_wide_heads_and_refuses_their_backward():
    import umfa_torch
    torch.manual_seed(1)
    q, k, v = (torch.randn(1, 2, 64, 384, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    from umfa_torch import sdpa
    before = sdpa.get_dispatch_stats()["pytorch_fallback"]
    o = umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert umfa_torch.last_kernel() == "fa_fwd_wide<512>"
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=True)
    assert float((o.float() - ref).abs().max()) < 2.0 ** -7 * float(ref.abs().max())
    assert sdpa.get_dispatch_stats()["pytorch_fallback"] == before
    # gradients: no kernel above head_dim 256 -- the surface falls back to torch (as the reference does for what its backward cannot serve)
    qg = q.clone().requires_grad_(True)
    og = umfa_torch.scaled_dot_product_attention(qg, k, v)
    og.float().sum().backward()
    assert qg.grad is not None and torch.isfinite(qg.grad).all() and sdpa.get_dispatch_stats()["pytorch_fallback"] == before + 1


def test_limits():
    """head_dim 1025 is refused (the reference: "Head dimension too large (max 1024)"), the backward stays at 256"""
    import umfa_torch
    from umfa._ffi import MFAError
    q, k, v = (torch.randn(1, 1, 8, 1032, device="cuda", dtype=torch.float16) for _ in range(3))
    with pytest.raises(MFAError) as ei:
        umfa_torch.attention_forward(q, k, v)
    assert ei.value.code == 1
