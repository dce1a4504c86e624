"""fp32 ADDITIVE mask tensors and the one-wave-per-SIMD bias kernels (end of round 6).  The reference's own callers build additive masks in fp32
(metal_sdpa_backend.cpp:3210-3231: bool -> fp32 0 / -inf, float -> fp32; semantics MFABridge.swift:157-242).  The bias kernels read fp16, so the
classification pass writes an fp16 copy AND decides on the device whether fp16 holds every value exactly; the call enqueues the bias kernel (on the copy)
and the 128-row kernel (on the caller's tensor), each guarded by that verdict (FwdParams::guard): exactly one runs.  What is pinned here:
  * an fp32 mask whose values fp16 holds gives, bit for bit, what the same values give as an fp16 tensor (the bias kernel ran);
  * an fp32 mask with a single value fp16 does not hold gives, bit for bit, what the 128-row kernel alone gives (option no_w64_f32_mask = 1);
  * both against the CPU oracle WITH the fp32 mask;
  * one captured graph follows the mask's CONTENTS from exact to inexact and back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tolerances import check_forward  # noqa: E402

NEG = float("-inf")


def _oracle():
    from oracle import oracle
    return oracle


def bits(t):
    return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def npy(t):
    return bits(t) if t.dtype == torch.bfloat16 else t.cpu().contiguous().numpy()


def _exact_mask(kind, B, H, Sq, Skv, seed):
    """fp32 tensors whose every value fp16 holds"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    i = torch.arange(Sq, device="cuda")[:, None]
    j = torch.arange(Skv, device="cuda")[None, :]
    if kind == "rel_pos_dyadic":               # [1, 1, Sq, Skv]: -|i - j| / 64: multiples of 2^-6 below 2^11 * 2^-6: exact
        return (-(i - j).abs().float() / 64.0)[None, None].contiguous()
    if kind == "bool_to_f32":                  # what the reference's torch path builds from a bool mask: 0 / -inf (one per batch element: a mask of its own per
        keep = torch.rand(B, 1, Sq, Skv, device="cuda", generator=g) < 0.7  # head would be past the route's size rule at these small shapes)
        keep[..., 3] = True
        return torch.where(keep, 0.0, NEG).contiguous()
    if kind == "blockdiag_inf":                # documents of 192 rows / 160 keys: open, masked and mixed tiles
        return torch.where((i // 192) == (j // 160), 0.0, NEG)[None, None].contiguous()
    if kind == "padding_row_broadcast":        # [B, 1, 1, Skv]
        lens = torch.tensor([max(1, Skv - 37 - 211 * b) for b in range(B)], device="cuda")
        return torch.where(j[None] < lens[:, None, None], 0.0, NEG)[:, None].contiguous()
    if kind == "widened_f16_random":           # a model's fp16 bias widened by .float(), 20 % -inf, per batch element
        m = (torch.randn(B, 1, Sq, Skv, device="cuda", generator=g) * 2.0).to(torch.float16)
        m[torch.rand(B, 1, Sq, Skv, device="cuda", generator=g) < 0.2] = NEG
        m[..., 5] = 0.5
        return m.float()
    if kind == "strided_rows":                 # the left half of a wider fp32 tensor
        wide = torch.randn(B, 1, Sq, 2 * Skv, device="cuda", generator=g).to(torch.float16).float()
        return wide[..., :Skv]
    if kind == "finfo_min":                    # torch.finfo(torch.float32).min where masked: its log2-domain term overflows to -inf in fp32 in every kernel of the
        # library (fa_common.h mask_term), so the pass copies it as -inf and calls it exact (every row keeps key block 0: the oracle, which works in fp64, agrees)
        return torch.where((i // 256) >= (j // 256), 0.0, torch.finfo(torch.float32).min)[None, None].contiguous()
    if kind == "tiny_values":                  # magnitudes below fp16's subnormals: flushed, e^x = 1 in fp32 either way
        return (torch.randn(1, 1, Sq, Skv, device="cuda", generator=g) * 1e-9).contiguous()
    raise ValueError(kind)


EXACT_KINDS = ["rel_pos_dyadic", "bool_to_f32", "blockdiag_inf", "padding_row_broadcast", "widened_f16_random", "strided_rows", "tiny_values", "finfo_min"]


@pytest.mark.parametrize("kind", EXACT_KINDS)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,grid", [((2, 2, 512, 512), 0), ((1, 6, 1280, 1408), 4)])
@pytest.mark.parametrize("D", [128, 64])
def test_fp32_mask_that_fp16_holds_runs_the_bias_kernel(kind, dt, shape, grid, D, umfa_opts):
    import umfa_torch
    umfa_opts(force_w64=1)
    if grid:
        umfa_opts(w64_grid=grid)
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    m32 = _exact_mask(kind, B, H, Sq, Skv, seed=Sq)
    assert m32.dtype == torch.float32
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m32, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert "bias> | fa_fwd16<" in kern and "chosen on the device" in kern, kern
    # the same values as an fp16 tensor: the bias kernel alone -- the guarded pair must give exactly that
    m16 = m32.to(torch.float16)
    if kind == "strided_rows":
        wide = torch.zeros(B, 1, Sq, 2 * Skv, device="cuda", dtype=torch.float16)
        wide[..., :Skv] = m16
        m16 = wide[..., :Skv]
    o16, lse16 = umfa_torch.attention_forward(q, k, v, mask=m16, out_dtype=torch.float32, return_lse=True)
    assert "bias>" in umfa_torch.last_kernel() and "|" not in umfa_torch.last_kernel()
    assert torch.equal(o, o16) and torch.equal(lse, lse16)
    mfull = np.ascontiguousarray(m32.expand(B, H, Sq, Skv).cpu().numpy())
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=mfull, mask_type=_oracle().MASK_ADDITIVE, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, dt, kern, f"w64_f32mask_{kind}", scale_max=1.02 if grid else 1.0)
    dead = np.isneginf(mfull).all(-1)
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert (on[dead] == 0).all() and np.isneginf(ln[dead]).all()
    assert np.abs(ln[~dead] - ref_lse[~dead]).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m32, out_dtype=torch.float32))  # repeatable
    # 16-bit output epilogue through the guarded pair
    ob = umfa_torch.attention_forward(q, k, v, mask=m32)
    assert ob.dtype == dt and float((ob.float() - o).abs().max()) <= 2.0 ** -8 * float(o.abs().max()) * 1.01


def _inexact_mask(kind, B, H, Sq, Skv, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    i = torch.arange(Sq, device="cuda")[:, None]
    j = torch.arange(Skv, device="cuda")[None, :]
    if kind == "rel_pos_thirds":               # -|i - j| / 3: not dyadic
        return (-(i - j).abs().float() / 3.0)[None, None].contiguous()
    if kind == "one_value":                    # block-diagonal 0 / -inf with ONE element fp16 does not hold, in the last tile of the last block
        m = torch.where((i // 192) == (j // 160), 0.0, NEG)[None, None].contiguous()
        m[0, 0, Sq - 1, Skv - 2] = 0.1
        return m
    if kind == "minus_1e9":                    # the "large negative" idiom at -1e9: finite in every domain, beyond fp16's range
        return torch.where((i // 256) >= (j // 256), 0.0, -1e9)[None, None].contiguous()
    if kind == "random_normal":                # N(0, 1) per batch element
        return torch.randn(B, 1, Sq, Skv, device="cuda", generator=g)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["rel_pos_thirds", "one_value", "minus_1e9", "random_normal"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,grid", [((2, 2, 512, 512), 0), ((1, 6, 1280, 1408), 4)])
@pytest.mark.parametrize("D", [128, 64])
def test_fp32_mask_that_fp16_does_not_hold_runs_the_128_row_kernel(kind, dt, shape, grid, D, umfa_opts):
    import umfa_torch
    umfa_opts(force_w64=1)
    if grid:
        umfa_opts(w64_grid=grid)
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv + 1)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    m32 = _inexact_mask(kind, B, H, Sq, Skv, seed=Sq)
    out = torch.full((B, H, Sq, D), float("nan"), device="cuda", dtype=torch.float32)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m32, out=out, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert "bias> | fa_fwd16<" in kern, kern
    with umfa_torch.options(no_w64_f32_mask=1):
        o1, lse1 = umfa_torch.attention_forward(q, k, v, mask=m32, out_dtype=torch.float32, return_lse=True)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    if dt == torch.float16:
        assert torch.equal(o, o1) and torch.equal(lse, lse1)
    else:
        # bf16 operands: the guarded launch takes the fp16 image of V the first route's cast pass wrote (V 2^-e, e from the slab), the lone one converts V in
        # the kernel at these sizes (e = 0) -- power-of-two scalings, the same numbers unless something underflows
        assert float((o - o1).abs().max()) <= 2.0 ** -12 * float(o1.abs().max()) and float((lse - lse1).abs().max()) <= 1e-5
    mfull = np.ascontiguousarray(m32.expand(B, H, Sq, Skv).cpu().numpy())
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=mfull, mask_type=_oracle().MASK_ADDITIVE, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, dt, kern.split(" | ")[1].split(" (")[0], f"w64_f32mask_inexact_{kind}")
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert np.abs(ln - ref_lse).max() < 2e-2


def test_fp32_mask_routing_conditions(umfa_opts):
    """by default (no force_w64) from one 256-row block per CU on; masks too large to read twice, unaligned rows, ragged shapes, causal + mask and the
    option keep the 128-row kernel alone"""
    import umfa_torch
    torch.manual_seed(5)
    B, H, S, D = 1, 72, 1024, 128  # 288 blocks >= 256 CUs
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    bias = (-(i[:, None] - i[None, :]).abs().float() / 128.0)[None, None].contiguous()
    o = umfa_torch.attention_forward(q, k, v, mask=bias, out_dtype=torch.float32)
    assert " | " in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    o16 = umfa_torch.attention_forward(q, k, v, mask=bias.to(torch.float16), out_dtype=torch.float32)
    assert torch.equal(o, o16)
    with umfa_torch.options(no_w64_f32_mask=1):
        o2 = umfa_torch.attention_forward(q, k, v, mask=bias, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<")
    assert float((o - o2).abs().max()) <= 2.0 ** -9 * float(o2.abs().max())
    umfa_torch.attention_forward(q, k, v, mask=bias, causal=True, out_dtype=torch.float32)
    assert " | " not in umfa_torch.last_kernel()
    wide = torch.zeros(1, 1, S, S + 2, device="cuda", dtype=torch.float32)
    wide[..., 2:] = bias
    ou = umfa_torch.attention_forward(q, k, v, mask=wide[..., 2:], out_dtype=torch.float32)  # rows start 8 bytes off a 16-byte boundary: read element by element by the pass
    assert " | " in umfa_torch.last_kernel() and torch.equal(ou, o)
    umfa_torch.attention_forward(q[:, :, :1000], k, v, mask=bias[:, :, :1000], out_dtype=torch.float32)  # Sq not a multiple of 64
    assert " | " not in umfa_torch.last_kernel()
    per_head = bias.expand(1, H, S, S).contiguous()  # 302 MB of mask against 2 x 94 MB of tensors: read once, by the 128-row kernel
    umfa_torch.attention_forward(q, k, v, mask=per_head, out_dtype=torch.float32)
    assert " | " not in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    del per_head
    # the size rule is shape-aware (fa_aux.hip mask_flags_worthwhile): 3.2 x the tensors' bytes is too much for a mask with a head dimension (a bias) and fine for one
    # with a batch dimension and none for the heads (a padding / document mask in additive form: up to 8 x) -- same bytes, same answers
    B2, H2, S2 = 4, 4, 4096  # 256 blocks; a [4,1,S,S] / [1,4,S,S] fp32 mask = 268 MB = 3.2 x the tensors
    q2, k2, v2 = (torch.randn(B2, H2, S2, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    j = torch.arange(S2, device="cuda")
    docs = torch.where((j[:, None] // 1024) == (j[None, :] // 1024), 0.0, NEG)
    o_b = umfa_torch.attention_forward(q2, k2, v2, mask=docs[None, None].expand(B2, 1, S2, S2).contiguous(), out_dtype=torch.float32)
    assert " | " in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    o_h = umfa_torch.attention_forward(q2, k2, v2, mask=docs[None, None].expand(1, H2, S2, S2).contiguous(), out_dtype=torch.float32)
    assert " | " not in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    assert float((o_b - o_h).abs().max()) <= 2.0 ** -9 * float(o_h.abs().max())


def test_fp32_mask_graph_replay_follows_the_contents_across_the_verdict():
    """ONE captured call; the fp32 mask's contents go exact -> inexact -> exact -> all -inf rows: the replay runs the kernel the contents ask for and equals the
    eager call bit for bit every time (no host read-back, no memset node, no state left behind by the route that was not taken)"""
    import umfa_torch
    torch.manual_seed(9)
    B, H, S, D = 1, 72, 1024, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    m = torch.zeros(1, 1, S, S, device="cuda", dtype=torch.float32)
    out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        umfa_torch.attention_forward(q, k, v, mask=m, out=out)
        assert " | " in umfa_torch.last_kernel()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            umfa_torch.attention_forward(q, k, v, mask=m, out=out)
    d = (i[:, None] - i[None, :]).abs().float()
    contents = [("exact", -d / 128.0), ("inexact", -d / 100.0), ("exact", torch.where((i[:, None] // 256) == (i[None, :] // 256), 0.0, NEG)),
                ("inexact", torch.where((i[:, None] // 256) == (i[None, :] // 256), 0.3, NEG)), ("exact", torch.zeros(S, S, device="cuda"))]
    for what, c in contents:
        m.copy_(c[None, None])
        out.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), what
        if what == "exact":
            want = umfa_torch.attention_forward(q, k, v, mask=m.to(torch.float16), out_dtype=torch.float32)
        else:
            with umfa_torch.options(no_w64_f32_mask=1):
                want = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        assert torch.equal(want, out), what
        assert torch.equal(umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32), out), what
    plain = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert float((plain - out).abs().max()) <= 2.0 ** -9 * float(plain.abs().max())  # the all-zero mask: the unmasked answer


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Sq", [320, 576, 1024])
@pytest.mark.parametrize("exact", [True, False, "bf16"])
def test_fp32_mask_blocks_that_see_nothing_and_a_ragged_last_block(dt, Sq, exact, umfa_opts):
    """the copy is written only where the bias kernel reads it (listed tiles; tile 0 of a block that sees nothing anywhere: its waves read their -inf there) -- a
    whole 256-row block at -inf, dead rows, Sq a multiple of 64 but not of 256 (found by the fuzz: seed 2, B2 H1 Sq320 Skv128 'dead', stale bytes read as a mask)"""
    import umfa_torch
    umfa_opts(force_w64=1)
    B, H, Skv, D = 2, 1, 128, 128
    g = torch.Generator(device="cuda").manual_seed(Sq)
    q = torch.randn(B, H, Sq, D, device="cuda", generator=g).to(dt)
    k = torch.randn(B, H, Skv, D, device="cuda", generator=g).to(dt)
    v = torch.randn(B, H, Skv, D, device="cuda", generator=g).to(dt)
    val = torch.randn(B, 1, Sq, Skv, device="cuda", generator=g) * 2.0
    if exact == "bf16":  # (the same rule for the fp16 copy of a bf16 mask, since the third session: written only where it is read)
        val = val.to(torch.bfloat16)
    elif exact:
        val = val.to(torch.float16).float()
    keep = torch.rand(B, 1, Sq, Skv, device="cuda", generator=g) < 0.6
    keep[:, :, ::5] = False
    keep[:, 0, 256:min(512, Sq)] = False
    m32 = val.masked_fill(~keep, NEG).contiguous()
    # poison the scratch the copy lives in: a call with a mask full of NaN first (same shapes: the same block)
    umfa_torch.attention_forward(q, k, v, mask=torch.full_like(m32, float("nan")), out_dtype=torch.float32)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m32, out_dtype=torch.float32, return_lse=True)
    assert (" | " in umfa_torch.last_kernel()) == (exact != "bf16") and "bias>" in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    assert torch.isfinite(o).all()
    s_ = torch.matmul(q.double(), k.double().transpose(-1, -2)) * D ** -0.5 + m32.double()
    rl = torch.logsumexp(s_, dim=-1)
    ref = torch.matmul(torch.nan_to_num(torch.softmax(s_, dim=-1), nan=0.0), v.double())
    assert float((o.double() - ref).abs().max() / ref.abs().max()) < 2.0 ** -11 * 1.5
    dead = ~torch.isfinite(rl)
    assert bool(dead.any())
    lg = lse.view(B, H, Sq)
    assert bool(torch.isneginf(lg[dead]).all()) and bool((o[dead.unsqueeze(-1).expand_as(o)] == 0).all())
    assert float((lg.double() - rl)[~dead].abs().max()) < 2e-2


def test_fp32_mask_call_captured_without_a_warm_up_runs_the_128_row_kernel_alone():
    """nothing is allocated under capture: a first capture on a stream the library has never seen has no scratch for the pass, so the call is the 128-row kernel
    alone (unguarded, the mask read in place); after an eager warm-up on the stream the same capture takes the guarded pair.  Both replay to the eager answer."""
    import umfa_torch
    torch.manual_seed(13)
    B, H, S, D = 1, 72, 1024, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    m = (-(i[:, None] - i[None, :]).abs().float() / 128.0)[None, None].contiguous()
    eager = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
    assert " | " in umfa_torch.last_kernel()
    out = torch.empty_like(eager)
    cold = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cold):
        with torch.cuda.graph(g, stream=cold):
            umfa_torch.attention_forward(q, k, v, mask=m, out=out)
            assert umfa_torch.last_kernel().startswith("fa_fwd16<") and " | " not in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    g.replay()
    torch.cuda.synchronize()
    assert float((out - eager).abs().max()) <= 2.0 ** -9 * float(eager.abs().max())
    warm = torch.cuda.Stream()
    with torch.cuda.stream(warm):
        umfa_torch.attention_forward(q, k, v, mask=m, out=out)
    warm.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(warm):
        with torch.cuda.graph(g2, stream=warm):
            umfa_torch.attention_forward(q, k, v, mask=m, out=out)
            assert " | " in umfa_torch.last_kernel()
    out.fill_(float("nan"))
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("mdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_finfo_min_masks_on_the_128_row_kernel_are_minus_inf_and_their_tiles_are_skipped(mdt, dt, umfa_opts):
    """the transformers idiom on a launch too small for the bias kernels (8 blocks): an additive mask 0 / torch.finfo(mask dtype).min, causal + padding per batch element.
    The term the 128-row kernel adds is value x log2 e = -inf in fp32 (fa_common.h mask_term), so these ARE -inf masks: bit for bit the answer of the -inf mask, rows with
    every key masked give O = 0 / LSE = -inf -- and since the third session of round 6 the tile-flag pass says so too (it compared raw bits with -inf's before and skipped
    nothing): bit-identical with and without the flags, as every flagged launch is."""
    import umfa_torch
    B, H, S, D = 2, 2, 512, 128
    torch.manual_seed(3)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
    i = torch.arange(S, device="cuda")
    lens = torch.tensor([S - 100, S - 300], device="cuda")
    keep = (i[None, :, None] >= i[None, None, :]) & (i[None, None, :] < lens[:, None, None]) & (i[None, :, None] < lens[:, None, None])
    m_min = torch.where(keep, 0.0, torch.finfo(mdt).min).to(mdt)[:, None].contiguous()
    m_inf = torch.where(keep, 0.0, NEG).to(mdt)[:, None].contiguous()
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m_min, out_dtype=torch.float32, return_lse=True)
    assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    o2, lse2 = umfa_torch.attention_forward(q, k, v, mask=m_inf, out_dtype=torch.float32, return_lse=True)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    with umfa_torch.options(no_mask_flags=1):
        o3, lse3 = umfa_torch.attention_forward(q, k, v, mask=m_min, out_dtype=torch.float32, return_lse=True)
    assert torch.equal(o, o3) and torch.equal(lse, lse3)
    de = (~keep.any(-1))[:, None, :].expand(B, H, S)
    assert bool(de.any()) and bool((o[de] == 0).all()) and bool(torch.isneginf(lse.view(B, H, S)[de]).all())
    s_ = torch.matmul(q.double(), k.double().transpose(-1, -2)) * D ** -0.5 + m_inf.double()
    ref = torch.matmul(torch.nan_to_num(torch.softmax(s_, dim=-1), nan=0.0), v.double())
    assert float((o.double() - ref).abs().max() / ref.abs().max()) < 2.0 ** -11 * 1.5


@pytest.mark.parametrize("mdt", [torch.bool, torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,causal", [((2, 3, 333, 777), False), ((1, 4, 500, 1001), False), ((2, 2, 257, 513), True), ((1, 2, 64, 130), False)])
@pytest.mark.parametrize("form", ["dense", "row_broadcast", "offset_view", "strided_keys"])
def test_128_row_kernel_reads_a_realigned_copy_of_masks_with_unaligned_rows(mdt, dt, shape, causal, form, umfa_opts):
    """fa_fwd16 reads a mask four keys at a time only when rows are contiguous and aligned to four elements and Skv is a multiple of four; anything else -- an odd length -- it
    read per score with scalar loads (B4 H16 S1111, fp16 bias: 464 us against ~130 at S 1112).  Since the very end of round 6 the runtime hands it a copy with rows padded to four
    keys (fa_aux.hip launch_mask_realign; -inf / false in the pad).  Same answers as the in-place read (option no_mask_realign) and as the oracle; rows that see nothing: O = 0."""
    import umfa_torch
    B, H, Sq, Skv = shape
    D = 128
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    g = torch.Generator(device="cuda").manual_seed(Skv)
    rows = 1 if form == "row_broadcast" else Sq
    keep = torch.rand(B, 1, rows, Skv, device="cuda", generator=g) < 0.7
    keep[..., 1] = True
    if rows > 1:
        keep[:, :, 5::11] = False  # rows that see nothing
    if mdt == torch.bool:
        m = keep
    else:
        m = (torch.randn(B, 1, rows, Skv, device="cuda", generator=g)).to(torch.float16).float().masked_fill(~keep, NEG).to(mdt)
    if form == "offset_view":  # rows start one element into a wider tensor
        wide = torch.zeros(B, 1, rows, Skv + 3, device="cuda", dtype=m.dtype)
        wide[..., 1:Skv + 1] = m
        m = wide[..., 1:Skv + 1]
    elif form == "strided_keys":  # every other element of a tensor twice as wide
        wide = torch.zeros(B, 1, rows, 2 * Skv, device="cuda", dtype=m.dtype)
        wide[..., ::2] = m
        m = wide[..., ::2]
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, causal=causal, out_dtype=torch.float32, return_lse=True)
    assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    with umfa_torch.options(no_mask_realign=1):
        o1, lse1 = umfa_torch.attention_forward(q, k, v, mask=m, causal=causal, out_dtype=torch.float32, return_lse=True)
    # (two code paths of the kernel -- four keys per load against one: the same terms, another contraction of score x scale + term; each is checked against fp64 below)
    assert float((o - o1).abs().max()) <= 2.0 ** -9 * float(o1.abs().max()) + 1e-30
    madd = torch.zeros(B, 1, rows, Skv, device="cuda", dtype=torch.float64).masked_fill(~m, NEG) if mdt == torch.bool else m.double()
    s_ = torch.matmul(q.double(), k.double().transpose(-1, -2)) * D ** -0.5 + madd
    if causal:
        s_ = s_.masked_fill(~torch.ones(Sq, Skv, dtype=torch.bool, device="cuda").tril(), NEG)
    rl = torch.logsumexp(s_, dim=-1)
    ref = torch.matmul(torch.nan_to_num(torch.softmax(s_, dim=-1), nan=0.0), v.double())
    assert torch.isfinite(o).all()
    assert float((o.double() - ref).abs().max() / ref.abs().max()) < 2.0 ** -11 * 1.5
    dead = ~torch.isfinite(rl)
    lg = lse.view(B, H, Sq)
    if bool(dead.any()):
        assert bool(torch.isneginf(lg[dead]).all()) and bool((o[dead.unsqueeze(-1).expand_as(o)] == 0).all())
    assert float((lg.double() - rl)[~dead].abs().max()) < 2e-2


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [128, 64])
def test_fp32_mask_of_an_odd_length_that_fp16_does_not_hold(dt, D, umfa_opts):
    """the guarded pair's second route on a mask with unaligned rows (Skv 1001): the 128-row kernel reads a realigned copy made by a launch that checks the same verdict
    (it read the caller's tensor per score before: S 4097, fp32 bias, ~1850 us).  Same answer as the 128-row kernel alone, and the oracle's."""
    import umfa_torch
    umfa_opts(force_w64=1)
    B, H, Sq, Skv = 2, 2, 1031, 1001
    torch.manual_seed(D)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    i = torch.arange(Sq, device="cuda")[:, None]
    j = torch.arange(Skv, device="cuda")[None, :]
    m = (-(i - j).abs().float() / 3.0)[None, None].contiguous()
    m[0, 0, 7::13] = NEG
    out = torch.full((B, H, Sq, D), float("nan"), device="cuda", dtype=torch.float32)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out=out, return_lse=True)
    assert " | " in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    with umfa_torch.options(no_w64_f32_mask=1):
        o1 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<")
    assert float((o - o1).abs().max()) <= 2.0 ** -12 * float(o1.abs().max())
    s_ = torch.matmul(q.double(), k.double().transpose(-1, -2)) * D ** -0.5 + m.double()
    rl = torch.logsumexp(s_, dim=-1)
    ref = torch.matmul(torch.nan_to_num(torch.softmax(s_, dim=-1), nan=0.0), v.double())
    assert torch.isfinite(o).all() and float((o.double() - ref).abs().max() / ref.abs().max()) < 2.0 ** -11 * 1.5
    dead = ~torch.isfinite(rl)
    assert bool(dead.any()) and bool((o[dead.unsqueeze(-1).expand_as(o)] == 0).all())
