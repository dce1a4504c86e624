"""Head dims 257 ... 1024 (fa_fwd_wide.hip, fa_bwd_wide.hip): the reference's callers admit head_dim <= 1024
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


def test_torch_sdpa_surface_takes_wide_heads_forward_and_backward():
    import umfa_torch
    torch.manual_seed(1)
    q, k, v = (torch.randn(1, 2, 64, 384, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    from umfa_torch import sdpa
    before = sdpa.get_dispatch_stats()["pytorch_fallback"]
    o = umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert umfa_torch.last_kernel() == "fa_fwd_wide<512>"
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=True)
    assert float((o.float() - ref).abs().max()) < 2.0 ** -7 * float(ref.abs().max())
    # gradients: fa_bwd_wide behind the same autograd Function as every other head dim (a torch fall-back until the backward existed)
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
    og = umfa_torch.scaled_dot_product_attention(qg, kg, vg, is_causal=True)
    w = torch.randn_like(og)
    (og.float() * w.float()).sum().backward()
    assert umfa_torch.last_kernel() == "fa_bwd_wide<512>", umfa_torch.last_kernel()
    assert sdpa.get_dispatch_stats()["pytorch_fallback"] == before
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    (torch.nn.functional.scaled_dot_product_attention(qr, kr, vr, is_causal=True) * w.float()).sum().backward()
    for g, r in ((qg.grad, qr.grad), (kg.grad, kr.grad), (vg.grad, vr.grad)):
        assert g.dtype == torch.bfloat16 and float((g.float() - r).abs().max()) < 2.0 ** -6 * float(r.abs().max())


@pytest.mark.parametrize("D", [320, 264, 512, 1024])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_fp32_host_arrays_through_the_blocking_abi(D, causal):
    """mfa_attention_backward above head_dim 256: the fp32-exact backward's bar (5e-5 of the largest gradient against the oracle's fp64), ragged
    sizes (neither a multiple of the 32-row tiles), D returned"""
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(D + causal)
    B, H, Sq, Skv = 1, 2, 70, 101
    q = rng.standard_normal((B, H, Sq, D)).astype(np.float32)
    k = rng.standard_normal((B, H, Skv, D)).astype(np.float32)
    v = rng.standard_normal((B, H, Skv, D)).astype(np.float32)
    do = rng.standard_normal((B, H, Sq, D)).astype(np.float32)
    o, lse = orc.sdpa_forward(q, k, v, causal=causal, return_lse=True)
    with umfa.MFAContext() as ctx:
        dq, dk, dv, dvec = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision="fp32", intermediate_precision="fp32", layout="bhsd")
        assert ctx.last_kernel == ("fa_bwd_wide<512>" if D <= 512 else "fa_bwd_wide<1024>"), ctx.last_kernel
    rdq, rdk, rdv, rd = orc.sdpa_backward(do, q, k, v, o, lse, causal=causal)
    for got, ref, name in ((dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv"), (dvec.reshape(rd.shape), rd, "D")):
        assert np.isfinite(got).all(), name
        assert float(np.abs(got - ref).max()) < 5e-5 * max(1.0, float(np.abs(ref).max())), (name, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("D,causal", [(320, True), (1024, False), (776, True)])
def test_backward_in_stream_entry(dtype, D, causal):
    """umfa_attention_backward_stream on device tensors of every operand type (fp32 arithmetic on the rounded operands; gradients fp32, cast by
    the caller), against the oracle on the same rounded inputs; bitwise repeatable"""
    import umfa_torch
    orc = _oracle()
    torch.manual_seed(D)
    B, H, Sq, Skv = 2, 2, 97, 130
    q, do = (torch.randn(B, H, Sq, D, device="cuda", dtype=dtype) for _ in range(2))
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=dtype) for _ in range(2))
    o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True)
    g1 = umfa_torch.attention_backward(do, q, k, v, o, lse, scale=D ** -0.5, causal=causal, keep_fp32=True)
    assert umfa_torch.last_kernel().startswith("fa_bwd_wide<"), umfa_torch.last_kernel()
    g2 = umfa_torch.attention_backward(do, q, k, v, o, lse, scale=D ** -0.5, causal=causal, keep_fp32=True)
    ref = orc.sdpa_backward(npy(do), npy(q), npy(k), npy(v), o.cpu().numpy(), lse.cpu().numpy().reshape(B, H, Sq), causal=causal)
    for a, b, r, name in zip(g1, g2, ref, ("dq", "dk", "dv")):
        assert torch.equal(a, b), name
        a = a.float().cpu().numpy()
        assert float(np.abs(a - r).max()) < 5e-5 * max(1.0, float(np.abs(r).max())), (name, float(np.abs(a - r).max()))


def test_limits():
    """head_dim 1025 is refused, forward and backward (the reference: "Head dimension too large (max 1024)")"""
    import umfa_torch
    from umfa._ffi import MFAError
    q, k, v = (torch.randn(1, 1, 8, 1032, device="cuda", dtype=torch.float16) for _ in range(3))
    with pytest.raises(MFAError) as ei:
        umfa_torch.attention_forward(q, k, v)
    assert ei.value.code == 1
    o = torch.zeros(1, 1, 8, 1032, device="cuda", dtype=torch.float32)
    lse = torch.zeros(8, device="cuda", dtype=torch.float32)
    with pytest.raises(MFAError) as ei:
        umfa_torch.attention_backward(q, q, k, v, o, lse, scale=1.0)
    assert ei.value.code == 1


# ---- round 6: the quantised entries and the fused-RoPE entry above head_dim 256 (round 5 returned MFA_ERROR_INVALID_ARGS there) ----
def _rel(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("D", [320, 512, 1024])
@pytest.mark.parametrize("bits,mode", [(8, "blockwise"), (8, "tensor"), (4, "blockwise")])
@pytest.mark.parametrize("causal", [False, True])
def test_quantized_forward_blocking_entry(D, bits, mode, causal):
    """mfa_quantized_forward_with_lse (MFABridge+Quantized.swift:227-358) at head dims 257 ... 1024: the quantiser's integers per 64-row block
    (two sweeps over the block), q * s as fp32 images, the wide fp32 forward -- against the oracle's quantised restatement"""
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(D + bits)
    shape = (1, 2, 150, D)
    q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    with umfa.MFAContext() as ctx:
        o, lse = umfa.quantized_attention(ctx, q, k, v, causal=causal, precision=f"int{bits}", quant_mode=mode, layout="bhsd", return_lse=True)
        assert ctx.last_kernel.startswith("fa_fwd_wide<i"), ctx.last_kernel
    ref, rlse = orc.quantized_forward(q, k, v, causal=causal, bits=bits, quant_mode=0 if mode == "tensor" else 2)
    assert np.isfinite(o).all()
    assert _rel(o, ref) < 2e-5, _rel(o, ref)  # same integers, fp32 arithmetic on both products: nothing rounded to 16 bits
    assert np.abs(lse.reshape(rlse.shape) - rlse).max() < 1e-4


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_quantized_forward_stream_entry_with_masks(dt):
    """in-stream, 16-bit operands, the ABI's dense fp32 mask and the caller's own bool tensor"""
    import umfa_torch
    orc = _oracle()
    torch.manual_seed(3)
    B, H, Sq, Skv, D = 1, 2, 130, 200, 384
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=dt) for _ in range(2))
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, bits=8, return_lse=True)
    assert umfa_torch.last_kernel().startswith("fa_fwd_wide<i8")
    ref, rlse = orc.quantized_forward(npy(q) if dt == torch.bfloat16 else q.cpu().numpy(), npy(k) if dt == torch.bfloat16 else k.cpu().numpy(),
                                      npy(v) if dt == torch.bfloat16 else v.cpu().numpy(), bits=8, quant_mode=2)
    assert _rel(o.cpu().numpy(), ref) < 2e-5
    keep = torch.rand(1, 1, Sq, Skv, device="cuda") > 0.3
    keep[..., 0] = True
    om = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=keep, bits=8)
    full = torch.zeros(B, H, Sq, Skv).masked_fill(~keep.cpu().expand(B, H, Sq, Skv), float("-inf")).numpy()
    refm, _ = orc.quantized_forward(npy(q) if dt == torch.bfloat16 else q.cpu().numpy(), npy(k) if dt == torch.bfloat16 else k.cpu().numpy(),
                                    npy(v) if dt == torch.bfloat16 else v.cpu().numpy(), mask=full, bits=8, quant_mode=2)
    assert _rel(om.cpu().numpy(), refm) < 2e-5


@pytest.mark.parametrize("D", [320, 640])
def test_quantized_backward_entries(D):
    """mfa_quantized_backward / umfa_quantized_backward_stream (MFABridge+Quantized.swift:365-533) above head_dim 256: quantise -> q * s as fp32 ->
    the wide fp32 backward; gradients of the de-quantised operands (STE) against the oracle's fp64 backward on the same operands"""
    import umfa_torch
    orc = _oracle()
    torch.manual_seed(D)
    B, H, S = 1, 2, 128
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32) for _ in range(4))
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, bits=8, return_lse=True)
    dq, dk, dv, status = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse, bits=8)
    assert umfa_torch.last_kernel().startswith("fa_bwd_wide"), umfa_torch.last_kernel()
    assert int(status.item()) == 0

    def fake(x):
        x = x.cpu().numpy()
        out = np.empty_like(x)
        for h in range(H):
            qv, sc = orc.quantize_symmetric(x[0, h], group=64 * D)
            out[0, h] = orc.dequantize(qv, sc, group=64 * D).reshape(S, D)
        return out
    fq, fk, fv = fake(q), fake(k), fake(v)
    rdq, rdk, rdv, _ = orc.sdpa_backward(do.cpu().numpy(), fq, fk, fv, o.cpu().numpy(), lse.cpu().numpy().reshape(B, H, S))
    for got, ref, name in [(dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")]:
        assert np.abs(got.cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), name


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [320, 1024])
def test_fused_rope_entry(dt, D):
    """umfa_rope_attention_forward_stream above head_dim 256: rotate Q and K (the rotate kernel), then the wide forward -- the reference's
    sequence (metal_sdpa_backend.cpp:1472-1641); bit-identical to the three calls, and against the oracle"""
    import umfa_torch
    from umfa_torch import ops
    from oracle import oracle
    torch.manual_seed(D)
    B, H, S = 2, 2, 96
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
    pos = torch.arange(S, device="cuda", dtype=torch.float32)[:, None]
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device="cuda", dtype=torch.float32) / D))
    ang = (pos * inv[None, :]).repeat_interleave(2, dim=1)
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    out = ops.rope_attention_forward(q, k, v, cos, sin, causal=True, out_dtype=torch.float32)
    assert umfa_torch.last_kernel().startswith("fa_fwd_wide<")
    ref3 = ops.attention_forward(ops.rope_rotate(q, cos, sin), ops.rope_rotate(k, cos, sin), v, causal=True, out_dtype=torch.float32)
    assert torch.equal(out, ref3)
    c, s_ = cos.cpu().numpy(), sin.cpu().numpy()
    if dt == torch.bfloat16:
        qr, kr = oracle.f32_to_bf16_bits(oracle.rope_rotate(npy(q), c, s_)), oracle.f32_to_bf16_bits(oracle.rope_rotate(npy(k), c, s_))
    else:
        qr, kr = oracle.rope_rotate(npy(q), c, s_), oracle.rope_rotate(npy(k), c, s_)
    ref = oracle.sdpa_forward(qr, kr, npy(v), causal=True)
    assert _rel(out.cpu().numpy(), ref) < (2e-5 if dt == torch.float32 else 1e-4)  # (fp32 sums over up to 1024 products of operands rounded to bf16 after the rotation)
