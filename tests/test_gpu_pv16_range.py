"""The default bf16 forward (bf16 Q K^T, fp16 P V) over bf16's WHOLE exponent range, on every entry and under hipGraph replay.

fp16 has five exponent bits where bf16 has eight.  The kernels therefore take V as V * 2^-e with one power of two e per (batch, KV head)
slab, chosen on the device from the slab's largest |v| -- by the cast pre-pass (fa_aux.hip cast_rows_bf16_f16_kernel: the one-wave-per-SIMD
kernels and long 128-row launches) or by the converting 128-row kernel itself (a workgroup whose outputs show that e = 0 did not do
sweeps its keys again) -- and give 2^e back in the epilogue.  What the reference does with such inputs is return finite, correct O from
mfa_attention_encode_mtl without ever synchronising (MFABridge.swift:2377-2543, metal_sdpa_backend.cpp:1308-1446); so: oracle parity
<= 1e-3 per slab for a 3e8 outlier, for V ~ 1e-6, for slabs of very different scale in one call -- in-stream, through the blocking ABI,
and inside a replayed graph whose V CHANGES between replays (nothing may be baked in at capture, nothing may be sticky)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NORTH_STAR = 1.0e-3


def _oracle():
    from oracle import oracle
    return oracle


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _regime(v, kind):
    """V of one regime, per head where that says more: 'outlier' one 3e8 value in head 0, 'tiny' all ~1e-6, 'huge' all ~1e30,
    'mixed' heads scaled 1e-6 / 1 / 1e4 / 1e-25 ..., 'sparse_tiny' ordinary rows with every third one scaled by 1e-20"""
    v = v.clone()
    H = v.shape[1]
    if kind == "plain":
        return v
    if kind == "outlier":
        v[0, 0, min(5, v.shape[2] - 1), 7] = 3.0e8
        return v
    if kind == "tiny":
        return (v.float() * 1e-6).to(torch.bfloat16)
    if kind == "huge":
        return (v.float() * 1e30).to(torch.bfloat16)
    if kind == "mixed":
        sc = [1e-6, 1.0, 1e4, 1e-25, 3e8, 1e-12, 1e20, 0.5]
        for h in range(H):
            v[:, h] = (v[:, h].float() * sc[h % len(sc)]).to(torch.bfloat16)
        return v
    if kind == "sparse_tiny":
        v[:, :, ::3] = (v[:, :, ::3].float() * 1e-20).to(torch.bfloat16)
        return v
    raise KeyError(kind)


def _per_slab_err(o, ref):
    """max over (batch, head) slabs of max|O - ref| / max|ref| of THAT slab (a slab with its own scale is judged against it)"""
    o = np.asarray(o, dtype=np.float64)
    worst = 0.0
    for b in range(ref.shape[0]):
        for h in range(ref.shape[1]):
            d = np.abs(ref[b, h]).max()
            if d == 0.0:
                assert np.abs(o[b, h]).max() == 0.0
                continue
            worst = max(worst, float(np.abs(o[b, h] - ref[b, h]).max() / d))
    return worst


REGIMES = ["plain", "outlier", "tiny", "huge", "mixed", "sparse_tiny"]


@pytest.mark.parametrize("kind", REGIMES)
@pytest.mark.parametrize("D,causal", [(128, False), (128, True), (64, True), (64, False), (256, False), (32, False)])
def test_converting_kernel_over_bf16_range(kind, D, causal):
    """the 128-row kernel with the conversion on V's way into LDS (short launches: no pre-pass) -- its own check + second sweep"""
    import umfa_torch
    torch.manual_seed(11 + D)
    B, H, Sq, Skv = 1, 4, 384, 448
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = _regime(torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16), kind)
    o = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32)
    kern = umfa_torch.last_kernel()
    assert kern.startswith("fa_fwd16<bf16,") and "pv16" in kern, kern
    assert torch.isfinite(o).all()
    ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v), causal=causal)
    e = _per_slab_err(o.cpu().numpy(), ref)
    assert e < NORTH_STAR, (kind, D, causal, e)
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32))  # decided from the data: repeatable


def test_converting_kernel_second_sweep_is_per_workgroup():
    """causal, the outlier late in the key range: q-blocks above it never stage its tile and keep their first sweep's bits; the rows
    that do see it are right"""
    import umfa_torch
    torch.manual_seed(5)
    q, k, v = (torch.randn(1, 2, 1024, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o0 = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
    assert umfa_torch.last_kernel() == "fa_fwd16<bf16,64,pv16>"
    vx = v.clone()
    vx[0, 1, 900, 3] = -7.0e9
    ox = umfa_torch.attention_forward(q, k, vx, causal=True, out_dtype=torch.float32)
    assert torch.isfinite(ox).all()
    assert torch.equal(ox[0, 0], o0[0, 0]) and torch.equal(ox[0, 1, :896], o0[0, 1, :896])  # (128-row q-blocks: rows < 896 never see key 900)
    ref = _oracle().sdpa_forward(bits(q), bits(k), bits(vx), causal=True)
    assert _per_slab_err(ox.cpu().numpy(), ref) < NORTH_STAR


@pytest.mark.parametrize("kind", ["outlier", "tiny", "mixed"])
def test_split_kv_parts_shift_on_their_own(kind):
    """decode-like launches: the key range of an item is cut into parts, each a workgroup that decides by itself; what they publish is
    in V's own scale, so the fold needs to know nothing"""
    import umfa_torch
    torch.manual_seed(7)
    B, H, Sq, Skv, D = 2, 8, 3, 8192, 128
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = _regime(torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16), kind)
    if kind == "outlier":
        v[1, 3, 5000, 100] = 2.0e10
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    assert umfa_torch.last_kernel() == "fa_fwd16<bf16,128,pv16,dec>"  # (three query rows: the decode form, four key quarters per 128-key tile)
    assert torch.isfinite(o).all()
    ref, ref_lse = _oracle().sdpa_forward(bits(q), bits(k), bits(v), return_lse=True)
    assert _per_slab_err(o.cpu().numpy(), ref) < NORTH_STAR
    assert np.abs(lse.cpu().numpy().reshape(ref_lse.shape) - ref_lse).max() < 2e-2


@pytest.mark.parametrize("kind", ["outlier", "tiny"])
def test_masks_windows_and_ragged_rows(kind):
    import umfa_torch
    torch.manual_seed(9)
    B, H, Sq, Skv, D = 1, 3, 200, 333, 128  # ragged: rows past Sq must not vote, a partial last key tile
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = _regime(torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16), kind)
    mask = torch.rand(1, 1, Sq, Skv, device="cuda") > 0.3
    mask[0, 0, 17] = False  # a row that sees nothing: O = 0
    for kw in ({"mask": mask}, {"window": (40, 25)}, {}):
        o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, **kw)
        assert "pv16" in umfa_torch.last_kernel()
        assert torch.isfinite(o).all()
        if "window" in kw:
            r = torch.arange(Sq, device="cuda")[:, None]
            c = torch.arange(Skv, device="cuda")[None, :]
            m = ((c >= r - 40) & (c <= r + 25))[None, None].cpu().numpy()
        else:
            m = kw["mask"].cpu().numpy() if "mask" in kw else None
        orc = _oracle()
        ref = orc.sdpa_forward(bits(q), bits(k), bits(v), mask=m, mask_type=orc.MASK_NONE if m is None else orc.MASK_BOOL)
        assert _per_slab_err(o.cpu().numpy(), ref) < NORTH_STAR, (kind, list(kw))


def test_non_finite_v_stays_non_finite():
    """inf / NaN in V are the caller's: the outputs they reach are non-finite (as in fp32 arithmetic), the rest of the call is untouched"""
    import umfa_torch
    torch.manual_seed(13)
    q, k, v = (torch.randn(1, 2, 256, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o0 = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    vx = v.clone()
    vx[0, 0, 3, 9] = float("inf")
    ox = umfa_torch.attention_forward(q, k, vx, out_dtype=torch.float32)
    assert not torch.isfinite(ox[0, 0, :, 9]).any() and torch.equal(ox[0, 1], o0[0, 1])
    with umfa_torch.options(force_w64=1):
        q2, k2, v2 = (torch.randn(1, 2, 512, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        v2[0, 1, 100, 0] = float("nan")
        o2 = umfa_torch.attention_forward(q2, k2, v2, out_dtype=torch.float32)
        assert "w64" in umfa_torch.last_kernel()
        assert torch.isnan(o2[0, 1, :, 0]).all() and torch.isfinite(o2[0, 0]).all() and torch.isfinite(o2[0, 1, :, 1:]).all()


@pytest.mark.parametrize("shape,kern", [((1, 32, 2048, 2048, 128), "fa_fwd16_w64<bf16,128,pv16>"), ((2, 4, 512, 512, 64), "fa_fwd16<bf16,64,pv16>")])
def test_graph_replay_follows_the_data(shape, kern):
    """ONE captured call, replayed with different V in the same tensor: ordinary -> 3e8 outlier -> 1e-6 -> mixed slabs -> ordinary again.
    Every replay inside 1e-3 of the oracle on its own data (round 4 baked the kernel choice in at capture and raised a status word for
    calls that came later), the first and the last bit-identical, the eager call on the same data bit-identical to the replay."""
    import umfa_torch
    B, H, Sq, Skv, D = shape
    torch.manual_seed(21)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v0 = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = v0.clone()
    out = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        umfa_torch.attention_forward(q, k, v, out=out)  # warm-up on the capture stream: its pool gets the scratch
        assert umfa_torch.last_kernel() == kern
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            umfa_torch.attention_forward(q, k, v, out=out)
    from oracle import parity
    rows = parity.sample_rows(Sq, groups=4)
    first = None
    for kind in ["plain", "outlier", "tiny", "mixed", "plain"]:
        v.copy_(_regime(v0, kind))
        out.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), kind
        ref = _oracle().sdpa_forward_rows(bits(q), bits(k), bits(v), rows)
        e = _per_slab_err(out[:, :, rows].cpu().numpy(), ref)
        assert e < NORTH_STAR, (kind, e)
        eager = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
        assert torch.equal(eager, out), kind
        if kind == "plain":
            if first is None:
                first = out.clone()
            else:
                assert torch.equal(first, out)


def test_reference_entries_over_bf16_range():
    """mfa_attention_encode_mtl (in-stream, fp32 O, never waits) and the blocking mfa_attention_forward on host arrays: the same answers"""
    import umfa
    import umfa_torch
    torch.manual_seed(31)
    B, H, S, D = 1, 8, 1536, 128
    q, k = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    v = _regime(torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16), "mixed")
    out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    umfa_torch.attention_encode(q, k, v, out)
    torch.cuda.synchronize()
    assert "pv16" in umfa_torch.last_kernel()
    from oracle import parity
    rows = parity.sample_rows(S, groups=4)
    ref = _oracle().sdpa_forward_rows(bits(q), bits(k), bits(v), rows)
    assert torch.isfinite(out).all() and _per_slab_err(out[:, :, rows].cpu().numpy(), ref) < NORTH_STAR
    with umfa.MFAContext() as ctx:
        oh = umfa.flash_attention_forward(ctx, bits(q), bits(k), bits(v), input_precision="bf16", intermediate_precision="bf16", layout="bhsd")
        assert "pv16" in ctx.last_kernel, ctx.last_kernel
    assert np.isfinite(oh).all() and _per_slab_err(oh[:, :, rows], ref) < NORTH_STAR


def test_broadcast_kv_heads_and_long_slabs():
    """zero-copy grouped-query views (V's head stride 0: ONE slab, one exponent, for all query heads) and slabs too long for the in-kernel
    amax exchange (more than 64 chunks: amax and cast as two launches)"""
    import umfa_torch
    torch.manual_seed(41)
    q = torch.randn(2, 8, 1024, 128, device="cuda", dtype=torch.bfloat16)
    k1, v1 = (torch.randn(2, 1, 1024, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    v1[1] = (v1[1].float() * 1e7).to(torch.bfloat16)
    v1[0, 0, 77, 5] = -1.0e-30  # (and something far below the slab's largest: rounds away, harmlessly)
    k, v = k1.expand(2, 8, 1024, 128), v1.expand(2, 8, 1024, 128)
    with umfa_torch.options(force_w64=1):
        o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
        assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128,pv16>"
    ref = _oracle().sdpa_forward(bits(q), bits(k.contiguous()), bits(v.contiguous()))
    assert _per_slab_err(o.cpu().numpy(), ref) < NORTH_STAR
    # S = 33024 keys: more than 64 chunks of 512 rows per slab
    ql = torch.randn(1, 2, 512, 128, device="cuda", dtype=torch.bfloat16)
    kl, vl = (torch.randn(1, 2, 33024, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    vl[0, 1] = (vl[0, 1].float() * 3e-9).to(torch.bfloat16)
    vl[0, 0, 33000, 64] = 5.0e11
    with umfa_torch.options(force_w64=1):
        ol = umfa_torch.attention_forward(ql, kl, vl, out_dtype=torch.float32)
        assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128,pv16>"
    refl = _oracle().sdpa_forward(bits(ql), bits(kl), bits(vl))
    assert torch.isfinite(ol).all() and _per_slab_err(ol.cpu().numpy(), refl) < NORTH_STAR


@pytest.mark.parametrize("shape", [(1, 24, 4096, 128), (2, 16, 2300, 64), (2, 2, 16384, 128)])
def test_cast_pass_workgroups_that_are_not_served_help_themselves(shape):
    """The slab exchange of the cast pre-pass is a bounded wait (option cast_wait_us): a workgroup whose slab mates do not show up in time
    -- they need not be resident: CU-masked streams, many streams at once -- reads the slab's amax itself.  With the bound at 0 every
    workgroup that arrives before the last of its slab does so: same exponent, same fp16 image, bit-identical O; and the exchange words are
    left clean for the next launch either way."""
    import umfa_torch
    B, H, S, D = shape
    torch.manual_seed(43)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    v = _regime(v, "mixed")
    v[0, 0, S - 1, D - 1] = -7.0e9
    outs = []
    for wait in (100, 0, 0, 100):
        with umfa_torch.options(force_w64=1, cast_wait_us=wait):
            outs.append(umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32))
            assert "w64<bf16" in umfa_torch.last_kernel() and "pv16" in umfa_torch.last_kernel()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    rows = slice(0, min(S, 512))
    ref = _oracle().sdpa_forward(bits(q[:, :, rows].contiguous()), bits(k), bits(v))
    assert _per_slab_err(outs[1][:, :, rows].cpu().numpy(), ref) < NORTH_STAR


def test_slab_exchanges_make_progress_on_a_stream_that_owns_four_cus():
    """hipExtStreamCreateWithCUMask, CUs 0 ... 3: a slab of the V cast pass has 64 workgroups here (16384 keys) and about 20 fit on four CUs at once, so the
    slab's amax exchange cannot complete by co-residency -- the bounded wait (cast_wait_us) has to, and does: no hang, and the bf16 forward and the
    quantised forward (the same exchange among the quantiser's V workgroups) equal the full-chip stream's results bit for bit."""
    import ctypes
    import umfa_torch
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        fn = hip.hipExtStreamCreateWithCUMask
    except (OSError, AttributeError):
        pytest.skip("no hipExtStreamCreateWithCUMask")
    stream = ctypes.c_void_p()
    mask = (ctypes.c_uint32 * 8)(0x0000000F, 0, 0, 0, 0, 0, 0, 0)
    if fn(ctypes.byref(stream), 8, mask) != 0:
        pytest.skip("CU-masked streams not available")
    ext = torch.cuda.ExternalStream(stream.value)
    torch.manual_seed(0)
    q = torch.randn(1, 2, 1024, 128, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(1, 2, 16384, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    v[0, 1] *= 1e-6
    with umfa_torch.options(force_w64=1):
        ref = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
        assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128,pv16>"
        qref = umfa_torch.quantized_attention_forward_stream(q, k, v)
        assert umfa_torch.last_kernel() == "fa_fwd_w64_i8<128>"
        torch.cuda.synchronize()
        with torch.cuda.stream(ext):
            for _ in range(2):
                o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
                oq = umfa_torch.quantized_attention_forward_stream(q, k, v)
                ext.synchronize()
                assert torch.equal(o, ref) and torch.equal(oq, qref)
    hip.hipStreamDestroy(stream)


def _row_rel_err(o, ref):
    """max over rows of max|O_row - ref_row| / max|ref_row| -- a row is judged against ITS OWN size (the per-slab metric above cannot see a row of
    ordinary size going wrong next to a row 1e12 times larger: slab-relative that error is 1e-12)"""
    o, ref = np.asarray(o, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    den = np.abs(ref).max(axis=-1)
    num = np.abs(o - ref).max(axis=-1)
    ok = den > 0
    return float((num[ok] / den[ok]).max())


@pytest.mark.parametrize("force_w64", [0, 1])
@pytest.mark.parametrize("log2_outlier", [12, 24])
def test_rows_beside_an_outlier_row_keep_their_own_precision(force_w64, log2_outlier):
    """In-slab dynamic range, ROW-relative (round-5 review, parity residue c).  One key's V row is 2^12 / 2^24 times the others, and the launch is causal
    with that key LAST: every query row but the last never attends to it and its exact O is ordinary.  The slab's ONE power of two puts the outlier
    at the top of fp16's range; rows 2^24 below it still sit in fp16's normal range (it spans 2^30), so every row keeps the north-star's tolerance against
    its own size."""
    import umfa_torch
    torch.manual_seed(77)
    B, H, S, D = 1, 2, 512, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    v[0, 0, S - 1] = (v[0, 0, S - 1].float() * 2.0 ** log2_outlier).to(torch.bfloat16)
    with umfa_torch.options(force_w64=force_w64):
        o = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
        kern = umfa_torch.last_kernel()
    assert ",pv16" in kern and ("w64" in kern) == bool(force_w64), kern
    ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v), causal=True)
    assert torch.isfinite(o).all()
    e = _row_rel_err(o.cpu().numpy(), ref)
    assert e < 2.0 * NORTH_STAR, (kern, log2_outlier, e)  # (row-relative: a row's own max, not the slab's; measured 3e-4 ... 6e-4)


@pytest.mark.parametrize("force_w64", [0, 1])
def test_documented_bound_of_the_per_slab_shift(force_w64):
    """... and the bound (INTEGRATION.md 'Range of V', include/umfa_abi.h): ONE power of two per (batch, KV head) slab cannot serve values more than ~2^29
    apart -- rows of V that far below their slab's largest value reach the fp16 product as subnormals or zero.  A 1e12 (2^40) row next to N(0, 1) rows: the
    slab-relative error stays inside 1e-3 and everything is finite, but query rows that attend ONLY to the ordinary rows come out near zero.  The
    documented remedy is the bf16 P V kernels (option pv_fp16 = 0: fp32's exponent range, 8-bit P): they return every row to bf16's own tolerance.
    This test pins BOTH halves, so that the limitation cannot change silently (a kernel that starts handling it will fail the first half: tighten it)."""
    import umfa_torch
    torch.manual_seed(78)
    B, H, S, D = 1, 2, 512, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    v[0, 0, S - 1] = (v[0, 0, S - 1].float() * 1e12).to(torch.bfloat16)
    ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v), causal=True)
    with umfa_torch.options(force_w64=force_w64):
        o = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
        kern = umfa_torch.last_kernel()
        assert ",pv16" in kern, kern
        assert torch.isfinite(o).all()
        assert _per_slab_err(o.cpu().numpy(), ref) < NORTH_STAR
        if force_w64:  # (the converting 128-row kernel shifts per WORKGROUP: only the workgroup that met the outlier loses its ordinary rows)
            assert _row_rel_err(o[:, 0:1].cpu().numpy(), ref[:, 0:1]) > 0.5      # head 0, the documented loss: ordinary rows flushed
        assert _row_rel_err(o[:, 1:2].cpu().numpy(), ref[:, 1:2]) < 2.0 * NORTH_STAR  # head 1: another slab, untouched
        with umfa_torch.options(pv_fp16=0):
            o8 = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
            assert ",pv16" not in umfa_torch.last_kernel()
        assert _row_rel_err(o8.cpu().numpy(), ref) < 1.5e-2  # bf16 P V: 8-bit P, every row against its own size
