"""quant_mode 3 (UMFA_QUANT_BLOCKWISE_FP8PV): int8 Q K^T + fp8 e4m3 P V on v_mfma_scale_f32_32x32x64_f8f6f4
(fa_fwd_w64_i8f8).  Opt-in fast mode: held against the oracle's restatement of ITS arithmetic (int8 block-wise Q / K, fp8 V
tiles, exact P) with a bound that is the measured cost of rounding P to e4m3, and reported against exact SDPA and against
the reference's int8 arithmetic so that the accuracy price is a number, not a claim."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _oracle():
    from oracle import oracle
    return oracle


def bits(t):
    return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


@pytest.fixture(autouse=True)
def _force_w64(umfa_opts):
    umfa_opts(force_w64=1)


@pytest.mark.parametrize("shape,causal", [((1, 2, 256, 256), False), ((1, 3, 512, 448), False), ((2, 2, 768, 768), True),
                                          ((1, 2, 1024, 1000), False), ((1, 1, 256, 65), True), ((1, 6, 2048, 2048), False)])
def test_fp8pv_vs_oracle_restatement(shape, causal):
    import umfa_torch
    from tolerances import errors, record
    orc = _oracle()
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16) * 1.7 + 0.3
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, quant_mode="blockwise_fp8pv", return_lse=True)
    torch.cuda.synchronize()
    assert umfa_torch.last_kernel() == "fa_fwd_w64_i8f8<128>", umfa_torch.last_kernel()
    assert torch.isfinite(o).all()
    got = o.cpu().numpy()
    ref = orc.quantized_forward_fp8pv(bits(q), bits(k), bits(v), causal=causal)             # the mode's arithmetic, exact P
    model = orc.quantized_forward_fp8pv(bits(q), bits(k), bits(v), causal=causal, p_fp8=True)  # + P rounded to e4m3
    e_ref, r_ref = errors(got, ref)
    e_mod, r_mod = errors(model, ref)
    record("fp8pv", shape=list(shape), causal=causal, max_vs_restatement=e_ref, rms_vs_restatement=r_ref,
           model_max=e_mod, model_rms=r_mod)
    # the kernel's distance from the exact-P restatement is what rounding P to 3 mantissa bits costs: at most 1.5 x the
    # statistical model's (the kernel rounds against a deferred reference max, the model against the exact one)
    assert r_ref < 1.5 * r_mod + 2e-3, (r_ref, r_mod)
    assert e_ref < 2.0 * e_mod + 5e-3, (e_ref, e_mod)
    # LSE: the row sum is taken over the ROUNDED P (what P V uses): within the fp8 rounding of a sum of many terms
    _, rl = orc.quantized_forward(bits(q), bits(k), bits(v), causal=causal)
    assert np.abs(lse.cpu().numpy().reshape(rl.shape) - rl).max() < 3e-2
    # bitwise reproducible (index-order fold of cut items included)
    o2 = umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, quant_mode="blockwise_fp8pv")
    assert torch.equal(o, o2)


def test_fp8pv_accuracy_price_is_reported_and_bounded():
    """N(0,1) operands, S = 2048: exact SDPA vs (a) the reference's int8 arithmetic (mode 2), (b) the fp8 P V mode."""
    import umfa_torch
    from tolerances import errors, record
    orc = _oracle()
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 4, 2048, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    exact = orc.sdpa_forward(bits(q), bits(k), bits(v))
    o2 = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise").cpu().numpy()
    o3 = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise_fp8pv").cpu().numpy()
    e2, r2 = errors(o2, exact)
    e3, r3 = errors(o3, exact)
    record("fp8pv_price", int8_max=e2, int8_rms=r2, fp8pv_max=e3, fp8pv_rms=r3)
    assert r2 < 2.2e-2 and r3 < 5.5e-2 and r3 < 3.2 * r2  # measured: 1.5e-2 vs 4.0e-2 (2.6 x)


def test_fp8pv_wide_dynamic_range_and_rescale():
    """outlier-dominated V tiles (per-tile power-of-two scales differ by 2^10) and rising scores (the deferred max moves,
    O AND the MFMA row sums are rescaled)"""
    import umfa_torch
    from tolerances import errors
    orc = _oracle()
    torch.manual_seed(5)
    B, H, Sq, Skv = 1, 2, 256, 1024
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    base = torch.randn(B, H, Skv, 128, device="cuda")
    ramp = torch.linspace(0.0, 1.0, Skv, device="cuda").view(1, 1, Skv, 1)
    direction = q.float().mean(dim=2, keepdim=True)
    direction = direction / direction.norm(dim=-1, keepdim=True)
    k = (base * 0.3 + ramp * 40.0 * direction * 11.3).to(torch.bfloat16)
    v = torch.randn(B, H, Skv, 128, device="cuda")
    v[:, :, 64:128] *= 1000.0
    v[:, :, 512:576] *= 1e-3
    v = v.to(torch.bfloat16)
    o = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise_fp8pv")
    assert umfa_torch.last_kernel() == "fa_fwd_w64_i8f8<128>" and torch.isfinite(o).all()
    ref = orc.quantized_forward_fp8pv(bits(q), bits(k), bits(v))
    e, r = errors(o.cpu().numpy(), ref)
    assert r < 8e-2 and e < 0.15, (e, r)  # few effective keys per row: single P roundings (2^-4) show


def test_fp8pv_falls_back_to_mode2_where_unsupported():
    import umfa_torch
    q, k, v = (torch.randn(1, 2, 128, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o3 = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise_fp8pv")
    assert umfa_torch.last_kernel() == "fa_fwd_i8<64>"
    o2 = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise")
    assert torch.equal(o2, o3)
