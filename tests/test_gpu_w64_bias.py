"""ADDITIVE fp16 / bf16 mask tensors on the one-wave-per-SIMD forward (fa_fwd16_w64<., 128 | 64, bias>, round 6): the kernel DMAs the wave's 64 x 64 mask tile
straight from the caller's tensor (any <= 4-D broadcastable fp16 tensor with 16-byte aligned rows) and adds mask / scale to the raw scores; a
classification pre-pass gives per-wave tile classes (all -inf: masked, all zero: open, else mixed) and the visited-tile list of every 256-row block,
so fully masked tiles are never staged.  Semantics are the reference's additive masks (MFABridge.swift:157-242: the value is added to the scaled
score; -inf masks; a row with every key at -inf gives O = 0, LSE = -inf), checked against the CPU oracle WITH the mask, against the 128-row kernel
(option no_w64_bias) and for run-to-run bitwise repeatability."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tolerances import check_forward  # noqa: E402


def _oracle():
    from oracle import oracle
    return oracle


def bits(t):
    return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def npy(t):
    return bits(t) if t.dtype == torch.bfloat16 else t.cpu().contiguous().numpy()


NEG = float("-inf")


def _bias(kind, B, H, Sq, Skv, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    i = torch.arange(Sq, device="cuda")[:, None]
    j = torch.arange(Skv, device="cuda")[None, :]
    if kind == "rel_pos":                      # [1, 1, Sq, Skv]: -|i - j| / 64, every tile mixed
        return (-(i - j).abs().float() / 64.0).to(torch.float16)[None, None].contiguous()
    if kind == "per_head":                     # [1, H, Sq, Skv]: ALiBi-like slopes
        sl = torch.tensor([2.0 ** -(h + 2) for h in range(H)], device="cuda")[:, None, None]
        return (-(i - j).abs().float()[None] * sl).to(torch.float16)[None].contiguous()
    if kind == "random":                       # [B, H, Sq, Skv]: N(0, 2) with 20 % -inf, a key always open
        m = (torch.randn(B, H, Sq, Skv, device="cuda", generator=g) * 2.0).to(torch.float16)
        m[torch.rand(B, H, Sq, Skv, device="cuda", generator=g) < 0.2] = NEG
        m[..., 5] = 0.5
        return m
    if kind == "blockdiag_inf":                # [1, 1, Sq, Skv]: 0 inside documents of 192 rows / 160 keys, -inf outside: open / masked / mixed tiles
        return torch.where((i // 192) == (j // 160), 0.0, NEG).to(torch.float16)[None, None].contiguous()
    if kind == "padding_row_broadcast":        # [B, 1, 1, Skv]: 0 / -inf per batch element, no row dimension
        lens = torch.tensor([max(1, Skv - 37 - 211 * b) for b in range(B)], device="cuda")
        return torch.where(j[None] < lens[:, None, None], 0.0, NEG).to(torch.float16)[:, None].contiguous()
    if kind == "all_zero":
        return torch.zeros(1, 1, Sq, Skv, dtype=torch.float16, device="cuda")
    if kind == "empty_rows_and_blocks":        # rows, a whole 256-row block and a whole head at -inf
        m = (torch.randn(B, H, Sq, Skv, device="cuda", generator=g)).to(torch.float16)
        m[:, :, 5::17] = NEG
        m[:, 0, 256:512] = NEG
        if H > 1:
            m[:, 1] = NEG
        return m
    if kind == "large_negative":               # -30000 instead of -inf (the "large negative" idiom): finite, exp underflows
        return torch.where((i // 256) >= (j // 256), 0.0, -30000.0).to(torch.float16)[None, None].contiguous()
    if kind == "strided_rows":                 # a view with a row stride of its own: the left half of a wider tensor (rows stay 16-byte aligned)
        wide = (torch.randn(B, 1, Sq, 2 * Skv, device="cuda", generator=g)).to(torch.float16)
        return wide[..., :Skv]
    raise ValueError(kind)


KINDS = ["rel_pos", "per_head", "random", "blockdiag_inf", "padding_row_broadcast", "all_zero", "empty_rows_and_blocks", "large_negative", "strided_rows"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dt,mdt", [(torch.bfloat16, torch.float16), (torch.float16, torch.float16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("shape,grid", [((2, 2, 512, 512), 0), ((1, 3, 1280, 768), 0), ((1, 3, 1280, 1408), 4), ((2, 2, 512, 512), 3)])
@pytest.mark.parametrize("D", [128, 64])
def test_w64_additive_mask_vs_oracle(kind, dt, mdt, shape, grid, D, umfa_opts):
    """mdt = bfloat16: the mask as a bf16 model has it -- the classification pass writes the fp16 copy the kernel reads (exact where fp16 holds the value)"""
    import umfa_torch
    umfa_opts(force_w64=1)
    if grid:
        umfa_opts(w64_grid=grid)  # few workgroups: blocks cut along their tile lists, parts folded
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    m = _bias(kind, B, H, Sq, Skv, seed=Sq)
    if mdt != torch.float16:
        m = m.to(mdt) if m.is_contiguous() else m.to(mdt)  # (same values rounded to bf16; a strided view stays a strided view of a wider tensor)
        if kind == "strided_rows":
            wide = torch.zeros(m.shape[0], 1, Sq, 2 * Skv, device="cuda", dtype=mdt)
            wide[..., :Skv] = m
            m = wide[..., :Skv]
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern in (f"fa_fwd16_w64<bf16,{D},pv16,bias>", f"fa_fwd16_w64<fp16,{D},bias>"), kern
    mfull = np.ascontiguousarray(m.expand(B, H, Sq, Skv).float().cpu().numpy())
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=mfull, mask_type=_oracle().MASK_ADDITIVE, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    # (cut blocks: the fold's fp32 recombination of a block's parts sits on top of the kernel's own error -- key padding at 264 of 512 keys with fp16
    # operands, three workgroups: 4.91e-4 against the 4.88e-4 of one fp16 ulp of P; 2 % of slack for the forced-grid cases, none for whole blocks)
    check_forward(on, ref, dt, kern, f"w64_bias_{kind}", scale_max=1.02 if grid else 1.0)
    dead = np.isneginf(mfull).all(-1)
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert (on[dead] == 0).all() and np.isneginf(ln[dead]).all()          # rows whose every key is at -inf: O = 0, LSE = -inf
    assert np.abs(ln[~dead] - ref_lse[~dead]).max() < 2e-2
    # bitwise repeatable, and the 128-row kernel's numbers class
    o2 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
    assert torch.equal(o, o2)
    with umfa_torch.options(no_w64_bias=1, force_w64=0, w64_grid=0):
        o3 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    assert float((o - o3).abs().max()) <= 2.0 ** -9 * float(o3.abs().max())
    # 16-bit output epilogue
    o16 = umfa_torch.attention_forward(q, k, v, mask=m)
    assert o16.dtype == dt and float((o16.float() - o).abs().max()) <= 2.0 ** -8 * float(o.abs().max()) * 1.01


def test_w64_additive_mask_routing_and_what_stays_on_the_128_row_kernel(umfa_opts):
    """the route's conditions (fwd_w64_supported): fp16 and bf16 masks with 16-byte aligned rows on whole tiles take the bias kernel by default from one
    256-row block per CU on (fp32 masks: as one of a guarded pair); ragged shapes, unaligned rows and causal + bias keep the 128-row kernel -- same answers either way"""
    import umfa_torch
    torch.manual_seed(5)
    B, H, S, D = 1, 72, 1024, 128  # 288 blocks >= 256 CUs
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    bias = (-(i[:, None] - i[None, :]).abs().float() / 128.0)[None, None]
    o = {}
    for name, m in (("f16", bias.to(torch.float16)), ("bf16", bias.to(torch.bfloat16)), ("f32", bias.clone())):
        o[name] = umfa_torch.attention_forward(q, k, v, mask=m.contiguous(), out_dtype=torch.float32)
        kern = umfa_torch.last_kernel()
        # (fp32: the guarded pair since the end of round 6 -- bias kernel on the fp16 copy | 128-row kernel on the fp32 tensor, the device picks;
        # -|i - j| / 128 below 1024 / 128 = 8 is a multiple of 2^-7: fp16 holds it, the bias kernel runs: tests/test_gpu_w64_f32_mask.py)
        assert "bias" in kern and (" | fa_fwd16<" in kern) == (name == "f32"), (name, kern)
    # (the three masks hold the same numbers up to their own rounding of -|i - j| / 128)
    assert float((o["f16"] - o["f32"]).abs().max()) < 2e-3 * float(o["f32"].abs().max())
    assert float((o["bf16"] - o["f32"]).abs().max()) < 1e-2 * float(o["f32"].abs().max())
    m16 = bias.to(torch.float16).contiguous()
    umfa_torch.attention_forward(q, k, v, mask=m16, causal=True, out_dtype=torch.float32)
    assert "bias" not in umfa_torch.last_kernel()
    wide = torch.zeros(1, 1, S, S + 4, device="cuda", dtype=torch.float16)
    wide[..., 4:] = m16
    ou = umfa_torch.attention_forward(q, k, v, mask=wide[..., 4:], out_dtype=torch.float32)  # rows start 8 bytes off a 16-byte boundary: the kernel cannot DMA them in place ...
    assert "bias" in umfa_torch.last_kernel()  # ... so (end of round 6) the pass reads them element by element into its padded copy; before: the 128-row kernel
    assert torch.equal(ou, o["f16"])
    with umfa_torch.options(no_w64_ragged_mask=1):
        umfa_torch.attention_forward(q, k, v, mask=wide[..., 4:], out_dtype=torch.float32)
        assert "bias" not in umfa_torch.last_kernel()
    umfa_torch.attention_forward(q[:, :, :1000], k, v, mask=m16[:, :, :1000], out_dtype=torch.float32)  # Sq not a multiple of 64
    assert "bias" not in umfa_torch.last_kernel()


def test_w64_additive_mask_replays_in_a_graph_and_follows_the_mask():
    """one captured call, the mask tensor's CONTENTS changed between replays (finite bias -> block-diagonal -inf -> all zero): classes, lists and the
    result follow the data; the replay equals the eager call bit for bit"""
    import umfa_torch
    torch.manual_seed(9)
    B, H, S, D = 1, 72, 1024, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    m = torch.zeros(1, 1, S, S, device="cuda", dtype=torch.float16)
    out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        umfa_torch.attention_forward(q, k, v, mask=m, out=out)
        assert "bias" in umfa_torch.last_kernel()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            umfa_torch.attention_forward(q, k, v, mask=m, out=out)
    contents = [(-(i[:, None] - i[None, :]).abs().float() / 100.0).to(torch.float16),
                torch.where((i[:, None] // 256) == (i[None, :] // 256), 0.0, NEG).to(torch.float16),
                torch.zeros(S, S, device="cuda", dtype=torch.float16)]
    for c in contents:
        m.copy_(c[None, None])
        out.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        eager = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        assert torch.isfinite(out).all() and torch.equal(eager, out)
    plain = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert float((plain - out).abs().max()) <= 2.0 ** -9 * float(plain.abs().max())  # the all-zero mask: the unmasked answer


@pytest.mark.parametrize("kind", ["random", "empty_rows_and_blocks", "per_head"])
def test_w64_additive_mask_too_large_to_classify(kind, umfa_opts):
    """a per-(batch, head) mask whose bytes exceed twice the call's Q + K + V + O traffic is not classified (the pass would cost what the attention costs):
    every wave-tile counts as mixed and every tile is listed -- -inf tiles, -inf rows and whole -inf blocks then go through the masking body instead of
    being skipped.  Same answers (oracle, O = 0 / LSE = -inf for rows that see nothing)."""
    import umfa_torch
    umfa_opts(force_w64=1)
    B, H, Sq, Skv, D = 1, 2, 1024, 4096, 128
    torch.manual_seed(3)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    m = _bias(kind, B, H, Sq, Skv, seed=5)
    assert m.numel() * 2 > 2 * (B * H * D * (Sq * 6 + 2 * Skv * 2))  # the rule (fa_aux.hip mask_flags_worthwhile)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern == "fa_fwd16_w64<bf16,128,pv16,bias>", kern
    mfull = np.ascontiguousarray(m.expand(B, H, Sq, Skv).float().cpu().numpy())
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=mfull, mask_type=_oracle().MASK_ADDITIVE, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, torch.bfloat16, kern, f"w64_bias_unclassified_{kind}")
    dead = np.isneginf(mfull).all(-1)
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert (on[dead] == 0).all() and np.isneginf(ln[dead]).all()
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32))


def test_bias_call_captured_without_a_warm_up_runs_on_the_kernel_that_needs_no_scratch():
    """nothing is allocated under capture (runtime_internal.h): an additive-mask call whose stream has no class / list scratch yet -- a first capture without a
    warm-up of that call -- goes to the 128-row kernel (mask read in place, V converted in the kernel); after an eager warm-up the same capture takes the bias
    kernel.  Both replay to the eager answer."""
    import umfa_torch
    torch.manual_seed(13)
    B, H, S, D = 1, 72, 1024, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    m = (-(i[:, None] - i[None, :]).abs().float() / 128.0).to(torch.float16)[None, None].contiguous()
    eager = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
    assert "bias" in umfa_torch.last_kernel()
    out = torch.empty_like(eager)
    cold = torch.cuda.Stream()  # a stream the library has never seen: empty pools
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cold):
        with torch.cuda.graph(g, stream=cold):
            umfa_torch.attention_forward(q, k, v, mask=m, out=out)
            assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
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
            assert "bias" in umfa_torch.last_kernel()
    out.fill_(float("nan"))
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("D", [128, 64])
def test_bf16_mask_with_finfo_min_is_minus_inf_on_both_routes(D, umfa_opts):
    """the transformers idiom: an additive bf16 mask 0 / torch.finfo(torch.bfloat16).min, causal + padding, one per batch element.  Its log2-domain term overflows to
    -inf in fp32, so the 128-row kernel (which reads the bf16 tensor itself) treats it as -inf: rows with every key masked give O = 0, LSE = -inf.  The bias route's
    fp16 copy clamped it to -65504 until the third session of round 6 (finite: such rows came out as softmax(s) V, and no tile was ever skipped); now it copies -inf:
    bit for bit what a -inf mask gives, and the 128-row kernel's answer."""
    import umfa_torch
    umfa_opts(force_w64=1)
    B, H, S = 2, 2, 512
    torch.manual_seed(D)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    lens = torch.tensor([S - 100, S - 300], device="cuda")
    keep = (i[None, :, None] >= i[None, None, :]) & (i[None, None, :] < lens[:, None, None]) & (i[None, :, None] < lens[:, None, None])  # causal + padding; padded QUERY rows see nothing
    m_min = torch.where(keep, 0.0, torch.finfo(torch.bfloat16).min).to(torch.bfloat16)[:, None].contiguous()
    m_inf = torch.where(keep, 0.0, NEG).to(torch.bfloat16)[:, None].contiguous()
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m_min, out_dtype=torch.float32, return_lse=True)
    assert "bias>" in umfa_torch.last_kernel()
    o2, lse2 = umfa_torch.attention_forward(q, k, v, mask=m_inf, out_dtype=torch.float32, return_lse=True)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    dead = ~keep.any(-1)  # [B, S]
    assert bool(dead.any())
    de = dead[:, None, :].expand(B, H, S)
    assert bool((o[de] == 0).all()) and bool(torch.isneginf(lse.view(B, H, S)[de]).all())
    with umfa_torch.options(no_w64_bias=1, force_w64=0):
        o3, lse3 = umfa_torch.attention_forward(q, k, v, mask=m_min, out_dtype=torch.float32, return_lse=True)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<")
    assert float((o - o3).abs().max()) <= 2.0 ** -9 * float(o3.abs().max())
    assert bool((o3[de] == 0).all()) and bool(torch.isneginf(lse3.view(B, H, S)[de]).all())


RAGGED = [(1, 3, 1288, 776), (2, 2, 1024, 1001), (1, 6, 1096, 2056), (2, 1, 2001, 75)]  # (Skv 1001, 75: rows that are not 16-byte aligned -- the pass reads them element by element)


@pytest.mark.parametrize("kind", ["rel_pos", "random", "blockdiag_inf", "padding_row_broadcast", "empty_rows_and_blocks", "all_zero"])
@pytest.mark.parametrize("dt,mdt", [(torch.bfloat16, torch.float16), (torch.float16, torch.float16), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float16, torch.float32)])
@pytest.mark.parametrize("shape,grid", [(RAGGED[0], 0), (RAGGED[1], 3), (RAGGED[2], 0), (RAGGED[3], 4)])
@pytest.mark.parametrize("D", [128, 64])
def test_w64_additive_mask_ragged_shapes_vs_oracle(kind, dt, mdt, shape, grid, D, umfa_opts):
    """(end of round 6) Sq from 1024 on and Skv that are not multiples of 64 (any Skv >= 64): the classification pass writes the fp16 copy
    PADDED to whole 64 x 64 tiles, keys past Skv and rows past Sq at -inf, and the bias kernel runs on that (fp32 masks: as one of the guarded pair).  Against the oracle with
    the mask, against the 128-row kernel (option no_w64_ragged_mask), repeatable; rows that see nothing give O = 0 / LSE = -inf."""
    import umfa_torch
    umfa_opts(force_w64=1)
    if grid:
        umfa_opts(w64_grid=grid)
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv + D)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    m = _bias(kind, B, 1, Sq, Skv, seed=Sq)  # (masks without a head dimension: a ragged mask must be small enough for the pass to read it -- it needs the padded copy)
    if mdt == torch.float32:
        m = m.float()  # (fp16-born values: fp16 holds them -- the bias kernel of the pair runs)
    elif mdt != torch.float16:
        m = m.to(mdt)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert "bias>" in kern and (" | " in kern) == (mdt == torch.float32), kern
    mfull = np.ascontiguousarray(m.expand(B, H, Sq, Skv).float().cpu().numpy())
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=mfull, mask_type=_oracle().MASK_ADDITIVE, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, dt, kern.split(" | ")[0], f"w64_bias_ragged_{kind}", scale_max=1.02 if grid else 1.0)
    dead = np.isneginf(mfull).all(-1)
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert (on[dead] == 0).all() and np.isneginf(ln[dead]).all()
    assert np.abs(ln[~dead] - ref_lse[~dead]).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32))
    with umfa_torch.options(no_w64_ragged_mask=1):
        o3 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    assert float((o - o3).abs().max()) <= 2.0 ** -9 * float(o3.abs().max())
    o16 = umfa_torch.attention_forward(q, k, v, mask=m)
    assert o16.dtype == dt and float((o16.float() - o).abs().max()) <= 2.0 ** -8 * float(o.abs().max()) * 1.01


def test_w64_additive_mask_ragged_routing(umfa_opts):
    """which ragged shapes take the bias kernels by default: Sq >= 1024, Skv >= 64 (any Skv: rows that are not 16-byte aligned are read element by element), a block per CU; the
    rest -- and the option -- the 128-row kernel"""
    import umfa_torch
    torch.manual_seed(5)
    B, H, D = 1, 72, 128
    for (Sq, Skv, mdt, want) in [(1000, 1000, torch.float16, False), (1096, 1000, torch.float16, True), (1096, 1000, torch.bfloat16, True), (1096, 1004, torch.float32, True),
                                 (1096, 1004, torch.float16, True), (1096, 1001, torch.float32, True), (1097, 1001, torch.bfloat16, True), (1096, 56, torch.float16, False)]:
        q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
        k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
        i = torch.arange(Sq, device="cuda")[:, None]
        j = torch.arange(Skv, device="cuda")[None, :]
        m = (-(i - j).abs().float() / 128.0).to(torch.float16).to(mdt)[None, None].contiguous()
        o = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        kern = umfa_torch.last_kernel()
        assert ("bias>" in kern) == want, (Sq, Skv, mdt, kern)
        with umfa_torch.options(no_w64_ragged_mask=1):
            o2 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
            assert "bias>" not in umfa_torch.last_kernel()
        assert float((o - o2).abs().max()) <= 2.0 ** -9 * float(o2.abs().max())
