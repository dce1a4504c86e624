"""GPU: F.scaled_dot_product_attention routed through the HIP kernels vs torch's own SDPA
(the reference's declared ground truth; tolerances conftest.py:186-199)."""
import numpy as np
import pytest
from tolerances import fam

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True)
def _clean():
    import umfa_torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a device")
    umfa_torch.reset_dispatch_stats()
    umfa_torch.set_quantization_mode(0, 0)
    yield
    umfa_torch.set_quantization_mode(0, 0)
    umfa_torch.unregister_backend()


TOL = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.5e-2}


def ref64(q, k, v, **kw):
    from umfa_torch import sdpa
    m = kw.pop("attn_mask", None)
    if m is not None and m.dtype != torch.bool:
        m = m.double()
    return sdpa._native_sdpa(q.double().cpu(), k.double().cpu(), v.double().cpu(),
                             attn_mask=None if m is None else m.cpu(), **kw).float()


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 12, 77, 64), (2, 4, 256, 64), (1, 24, 1536, 128)])
def test_inference_matches_torch(dt, shape):
    import umfa_torch
    torch.manual_seed(42)
    q, k, v = (torch.randn(shape, device="cuda", dtype=dt) * 0.1 for _ in range(3))  # conftest.py:149-158
    for causal in (False, True):
        out = umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=causal)
        assert out.dtype == dt and out.shape == q.shape  # test_scale_factor_fix.py:112-114
        assert (out.float().cpu() - ref64(q, k, v, is_causal=causal)).abs().max() < TOL[dt]
    s = umfa_torch.get_dispatch_stats()
    assert s["fp32_instream"] == 2 and s["pytorch_fallback"] == 0


def test_default_scale_and_promotion():
    import umfa_torch
    q, k, v = (torch.randn(32, 32, device="cuda") for _ in range(3))
    a = umfa_torch.scaled_dot_product_attention(q, k, v)
    b = umfa_torch.scaled_dot_product_attention(q, k, v, scale=1.0 / np.sqrt(32))
    assert a.shape == (32, 32) and (a - b).abs().max() < 1e-7  # test_scale_factor_fix.py:68-98


def test_gqa_and_masks():
    import umfa_torch
    torch.manual_seed(1)
    q = torch.randn(2, 8, 128, 64, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(2, 2, 128, 64, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(2, 2, 128, 64, device="cuda", dtype=torch.bfloat16)
    out = umfa_torch.scaled_dot_product_attention(q, k, v, enable_gqa=True)
    ref = ref64(q, k.repeat_interleave(4, 1), v.repeat_interleave(4, 1))
    assert (out.float().cpu() - ref).abs().max() < 2e-2
    kk, vv = k.repeat_interleave(4, 1), v.repeat_interleave(4, 1)
    mb = torch.rand(2, 1, 128, 128, device="cuda") > 0.2
    mb[..., 0] = True
    out = umfa_torch.scaled_dot_product_attention(q, kk, vv, attn_mask=mb)
    assert (out.float().cpu() - ref64(q, kk, vv, attn_mask=mb)).abs().max() < 2e-2
    ma = torch.randn(128, 128, device="cuda", dtype=torch.bfloat16)
    out = umfa_torch.scaled_dot_product_attention(q, kk, vv, attn_mask=ma)
    assert (out.float().cpu() - ref64(q, kk, vv, attn_mask=ma.float())).abs().max() < 2e-2
    # all-true bool mask == no mask (vi)
    allt = torch.ones(128, 128, dtype=torch.bool, device="cuda")
    a = umfa_torch.scaled_dot_product_attention(q, kk, vv, attn_mask=allt)
    b = umfa_torch.scaled_dot_product_attention(q, kk, vv)
    # the mask variant scales then subtracts, the plain one fuses both in one fma: equal to a bf16 ulp
    assert torch.allclose(a.float(), b.float(), rtol=2 ** -7, atol=1e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
def test_autograd_matches_torch(dt, causal):
    import umfa_torch
    torch.manual_seed(3)
    shape = (1, 4, 160, 64)
    q, k, v = (torch.randn(shape, device="cuda", dtype=dt, requires_grad=True) for _ in range(3))
    do = torch.randn(shape, device="cuda", dtype=dt)
    out = umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=causal)
    out.backward(do)
    got = [t.grad.float().cpu() for t in (q, k, v)]
    q2, k2, v2 = (t.detach().double().cpu().requires_grad_(True) for t in (q, k, v))
    from umfa_torch import sdpa
    sdpa._native_sdpa(q2, k2, v2, is_causal=causal).backward(do.double().cpu())
    tol = 2e-4 if dt == torch.float32 else 6e-2
    for g, r in zip(got, (q2.grad, k2.grad, v2.grad)):
        assert (g - r.float()).abs().max() < tol * max(1.0, float(r.abs().max()))
    assert umfa_torch.get_dispatch_stats()["fp32_autograd"] == 1


def test_quantized_mode_routes_and_trains():
    import umfa_torch
    torch.manual_seed(4)
    umfa_torch.set_quantization_mode(umfa_torch.QUANT_INT8, umfa_torch.QUANT_BLOCK_WISE)
    q, k, v = (torch.randn(1, 2, 128, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    out = umfa_torch.scaled_dot_product_attention(q, k, v)
    assert out.dtype == torch.bfloat16
    ref = ref64(q.detach(), k.detach(), v.detach())
    assert (out.float().cpu() - ref).abs().max() / ref.abs().max() < 0.08
    out.float().sum().backward()
    assert all(torch.isfinite(t.grad).all() for t in (q, k, v))
    s = umfa_torch.get_dispatch_stats()
    assert s["quantized_autograd"] == 1


def test_functional_patch_flux_style():
    # examples/flux: monkey-patch F.scaled_dot_product_attention, run, restore
    import torch.nn.functional as F
    import umfa_torch
    q, k, v = (torch.randn(1, 24, 512, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    with umfa_torch.use_umfa_sdpa():
        out = F.scaled_dot_product_attention(q, k, v)
    assert fam(umfa_torch.last_kernel()) in ("fa_fwd16<bf16,128>", "fa_fwd16_w64<bf16,128>")
    assert (out.float().cpu() - ref64(q, k, v)).abs().max() < 3e-2  # test_integration_flux.py:93-95 uses 0.1


def _rope_tables(S, D, B=None):
    g = torch.Generator().manual_seed(5)
    ang = torch.rand((S, D // 2) if B is None else (B, S, D // 2), generator=g) * 6.283
    return ang.cos().repeat_interleave(2, -1).cuda(), ang.sin().repeat_interleave(2, -1).cuda()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_rope_sdpa_matches_eager(dt):
    import umfa_torch
    from umfa_torch import sdpa
    torch.manual_seed(6)
    B, H, S, D = 2, 4, 192, 64
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
    for batched in (False, True):
        cos, sin = _rope_tables(S, D, B if batched else None)
        out = umfa_torch.rope_scaled_dot_product_attention(q, k, v, cos, sin, is_causal=True)
        ref = ref64(sdpa.apply_rope_eager_bhsd(q.float(), cos, sin), sdpa.apply_rope_eager_bhsd(k.float(), cos, sin), v,
                    is_causal=True)
        assert (out.float().cpu() - ref).abs().max() < (2e-5 if dt == torch.float32 else 3e-2)
    assert umfa_torch.get_dispatch_stats()["rope_instream"] == 2


def test_rope_sdpa_autograd():
    import umfa_torch
    from umfa_torch import sdpa
    torch.manual_seed(7)
    B, H, S, D = 1, 2, 96, 64
    q, k, v = (torch.randn(B, H, S, D, device="cuda", requires_grad=True) for _ in range(3))
    cos, sin = _rope_tables(S, D)
    do = torch.randn(B, H, S, D, device="cuda")
    umfa_torch.rope_scaled_dot_product_attention(q, k, v, cos, sin).backward(do)
    got = [t.grad.clone() for t in (q, k, v)]
    q2, k2, v2 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    cb, sb = cos.double().unsqueeze(0).unsqueeze(0), sin.double().unsqueeze(0).unsqueeze(0)

    def rope64(x):
        pairs = x.reshape(*x.shape[:-1], D // 2, 2)
        rot = torch.stack((-pairs[..., 1], pairs[..., 0]), -1).reshape(x.shape)
        return x * cb + rot * sb
    sdpa._native_sdpa(rope64(q2), rope64(k2), v2).backward(do.double())
    for g, r in zip(got, (q2.grad, k2.grad, v2.grad)):
        assert (g.double() - r).abs().max() < 2e-4 * max(1.0, float(r.abs().max()))
    s = umfa_torch.get_dispatch_stats()
    assert s["rope_autograd"] == 1


@pytest.mark.parametrize("shape", [(2, 8, 2, 128, 64), (1, 32, 8, 2048, 128), (3, 6, 3, 200, 80)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gqa_zero_copy_views(shape, dt):
    """inference GQA: K / V are NOT expanded -- the query heads of a KV head become the heads of a (batch x kv-head)
    slab with K / V head stride 0 on the in-stream entry; same numbers as the repeat_interleave route"""
    import umfa_torch
    from umfa_torch import sdpa
    B, Hq, Hkv, S, D = shape
    torch.manual_seed(5)
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt)
    v = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt)
    g = Hq // Hkv
    for causal, mask in ((False, None), (True, None), (False, (torch.rand(S, S, device="cuda") < 0.8) | torch.eye(S, dtype=torch.bool, device="cuda"))):
        assert sdpa._gqa_zero_copy(q, k, v, mask, 0.0, causal, None) is not None
        out = umfa_torch.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=causal, enable_gqa=True)
        exp = umfa_torch.scaled_dot_product_attention(q, k.repeat_interleave(g, 1), v.repeat_interleave(g, 1), attn_mask=mask,
                                                      is_causal=causal)
        assert out.shape == q.shape and out.dtype == dt
        assert (out.float() - exp.float()).abs().max().item() < (2e-2 if dt == torch.bfloat16 else 4e-3)
    # per-head masks and gradients keep the expanding route
    assert sdpa._gqa_zero_copy(q, k, v, torch.ones(B, Hq, S, S, dtype=torch.bool, device="cuda"), 0.0, False, None) is None
    assert sdpa._gqa_zero_copy(q.clone().requires_grad_(True), k, v, None, 0.0, False, None) is None


@pytest.mark.parametrize("shape", [(2, 32, 8, 1, 1000, 128), (1, 16, 2, 4, 777, 64), (3, 8, 1, 16, 300, 128), (2, 6, 3, 1, 130, 80), (1, 64, 8, 2, 4096, 128)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gqa_decode_packs_query_heads_into_rows(shape, dt, monkeypatch):
    """decode-like grouped-query calls (round 6): the g query heads of a KV head become ROWS of one 128-row tile ([B Hkv, 1, g Sq, D] is a plain view of a
    head-major q): K / V of a KV head are staged once instead of once per query head.  Same numbers as the head-view route and as torch on the
    expanded tensors; masks, causal, more than 128 rows per KV head and a q whose heads are not head-major keep the head views"""
    import umfa_torch
    B, Hq, Hkv, Sq, Skv, D = shape
    g = Hq // Hkv
    torch.manual_seed(Hq + Skv)
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt)
    k, v = (torch.randn(B, Hkv, Skv, D, device="cuda", dtype=dt) for _ in range(2))
    def ref_of(q_, keep=None):  # fp64, written out (torch's own GPU SDPA is not a reference: 8e-3 off on masked fp32 decode shapes here)
        s_ = (q_.double() @ k.double().repeat_interleave(g, 1).transpose(-1, -2)) * D ** -0.5
        if keep is not None:
            s_ = s_.masked_fill(~keep, float("-inf"))
        return (torch.softmax(s_, -1) @ v.double().repeat_interleave(g, 1)).float()
    ref = ref_of(q)
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    monkeypatch.setenv("UMFA_GQA_PACK_ROWS", "1")
    umfa_torch.reset_dispatch_stats()
    out = umfa_torch.scaled_dot_product_attention(q, k, v, enable_gqa=True)
    assert umfa_torch.get_dispatch_stats()["fp32_instream"] == 1
    assert out.shape == q.shape and out.dtype == dt
    assert (out.float() - ref).abs().max().item() < tol
    monkeypatch.setenv("UMFA_GQA_PACK_ROWS", "0")
    views = umfa_torch.scaled_dot_product_attention(q, k, v, enable_gqa=True)
    assert (out.float() - views.float()).abs().max().item() < tol / 4  # (rows of one tile instead of heads of a slab: same arithmetic per row)
    monkeypatch.setenv("UMFA_GQA_PACK_ROWS", "1")
    # a q that is not head-major (BSHD storage viewed as BHSD): the head views
    qs = torch.randn(B, Sq, Hq, D, device="cuda", dtype=dt).transpose(1, 2)
    outs = umfa_torch.scaled_dot_product_attention(qs, k, v, enable_gqa=True)
    refs = ref_of(qs)
    assert (outs.float() - refs).abs().max().item() < tol
    # a key-padding mask: the head views (rows of different heads would need the mask re-viewed)
    keep = (torch.arange(Skv, device="cuda") < Skv - 7)[None, None, None, :]
    outm = umfa_torch.scaled_dot_product_attention(q, k, v, attn_mask=keep, enable_gqa=True)
    refm = ref_of(q, keep)
    assert (outm.float() - refm).abs().max().item() < tol


@pytest.mark.parametrize("shape", [(2, 8, 2, 256, 64), (1, 32, 8, 1024, 128), (1, 4, 1, 320, 128), (1, 6, 3, 200, 80)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
def test_gqa_training_reads_k_v_in_place(shape, dt, causal):
    """training GQA: no repeat_interleave copies in forward or backward (umfa_attention_backward_gqa_stream); O and the
    gradients equal the expanding route's -- dQ bit for bit (same kernels, same operands), dK / dV up to the order of
    the group sum (fp32 sum in the library vs torch's sum of rounded per-head gradients)"""
    import umfa_torch
    B, Hq, Hkv, S, D = shape
    g = Hq // Hkv
    torch.manual_seed(9)
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt, requires_grad=True)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt, requires_grad=True)
    v = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt, requires_grad=True)
    do = torch.randn(B, Hq, S, D, device="cuda", dtype=dt)
    # (one forward kernel family for both routes: the dispatcher's cost model prices the V cast pass by DISTINCT V slabs -- 8 here, 32 on the expanding
    # route -- and may send the two to different structures; the bit-for-bit statement below is about the backward)
    with umfa_torch.options(no_w64=1):
        out = umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=True)
        out.backward(do)
        kern = umfa_torch.last_kernel()
        gq, gk, gv = q.grad.clone(), k.grad.clone(), v.grad.clone()
        q2, k2, v2 = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        ke, ve = k2.repeat_interleave(g, 1), v2.repeat_interleave(g, 1)
        exp = umfa_torch.scaled_dot_product_attention(q2, ke, ve, is_causal=causal)
        exp.backward(do)
    tol = 2e-2 if dt == torch.bfloat16 else 4e-3
    assert (out.float() - exp.float()).abs().max().item() < tol
    if D in (64, 128, 256):
        assert kern.startswith("fa_bwd16"), kern
        assert torch.equal(gq, q2.grad)
    for a, b_ in ((gq, q2.grad), (gk, k2.grad), (gv, v2.grad)):
        assert a.shape == b_.shape
        assert (a.float() - b_.float()).abs().max().item() <= tol * max(1.0, b_.float().abs().max().item())
