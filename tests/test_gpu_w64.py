"""GPU parity of the 64-rows-per-wave persistent forward (fa_fwd16_w64, head_dim 128 and 64, no mask, non-causal,
Sq % 256 == 0, Skv % 64 == 0) against the CPU oracle, through the in-stream C ABI.  The shapes are chosen to
cover: one-tile and odd/even tile counts (the two score sets swap roles), items cut into many parts by the
slice boundaries (small grids: every workgroup gets one tile), whole items, the mix of both (FLUX), strided
inputs, fp16, fp32 and 16-bit outputs, LSE, run-to-run bitwise determinism and the deferred-max rescale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _oracle():
    from oracle import oracle
    return oracle


@pytest.fixture(autouse=True)
def _force_w64(request, umfa_opts):
    """The dispatcher sends shapes with too little parallel work for one workgroup per CU to the 128-row kernel
    (fa_fwd16_w64.hip: fwd_w64_supported); the parity cases here are small on purpose, so lift that gate
    (umfa_set_option, restored after the test)."""
    if "dispatch_gate" not in request.node.name:
        umfa_opts(force_w64=1)


def bits(t):
    return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def npy(t):
    return bits(t) if t.dtype == torch.bfloat16 else t.cpu().contiguous().numpy()


def rel_err(a, ref):
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


from tolerances import check_forward, fam  # noqa: E402  (measured bounds, tests/tolerances.py)


def _band(Sq, Skv, window, causal):
    i = np.arange(Sq)[:, None]
    j = np.arange(Skv)[None, :]
    keep = (j >= i - window[0]) & (j <= i + window[1])
    return keep & (j <= i) if causal else keep


@pytest.mark.parametrize("shape", [(1, 2, 256, 64), (1, 2, 256, 128), (1, 1, 256, 192), (2, 3, 512, 256), (1, 2, 256, 1024),
                                   (1, 5, 768, 448)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_w64_small_shapes_vs_oracle(shape, dt):
    import umfa_torch
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64"), umfa_torch.last_kernel()
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), return_lse=True)
    check_forward(o.cpu().numpy(), ref, dt, umfa_torch.last_kernel(), inputs=(npy(q), npy(k), npy(v)))
    assert np.abs(lse.cpu().numpy().reshape(ref_lse.shape) - ref_lse).max() < 2e-2
    # 16-bit epilogue = the fp32 result rounded once (half an ulp of the 16-bit type)
    o16 = umfa_torch.attention_forward(q, k, v)
    assert o16.dtype == dt
    assert (o16.float() - o).abs().max() <= (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * o.abs().max() * 1.01
    # bitwise reproducible, including the index-order fold of split items
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32))


def test_w64_flux_shape_matches_oracle_rows_and_is_deterministic():
    import umfa_torch
    torch.manual_seed(0)
    B, H, S, D = 1, 24, 4096, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    assert fam(umfa_torch.last_kernel()) == "fa_fwd16_w64<bf16,128>"
    assert torch.isfinite(o).all()
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32))
    orc = _oracle()
    # every head (whole items, and items cut by a slice boundary at this shape: 1.5 items per workgroup), rows on and off
    # the 64 / 256-row grid, all keys; the oracle's row-subset entry keeps the whole-tensor normalisation of the metric
    rows = np.array([0, 255, 256, 300, 511, 512, 1000, 2047, 2048, 3333, 4095] + list(range(1536, 1600)))
    ref, rl = orc.sdpa_forward_rows(bits(q), bits(k), bits(v), rows, return_lse=True)
    inp = (bits(q), bits(k), bits(v))
    check_forward(o[:, :, rows].cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), "flux_rows", inputs=inp, rows=rows)
    assert np.abs(lse.view(B, H, S)[:, :, rows].cpu().numpy() - rl).max() < 2e-2
    # the other regimes of the same kernel: exact running max (sits at the bf16 format floor) and the deferred max
    for regime in ({"softmax_reference": "exact"}, {"softmax_reference": "deferred", "softmax_tau": 6}):
        with umfa_torch.options(**regime):
            o0 = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
            check_forward(o0[:, :, rows].cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(),
                          "flux_rows_" + regime["softmax_reference"], inputs=inp, rows=rows)


def test_w64_strided_inputs_and_cross_attention():
    import umfa_torch
    torch.manual_seed(3)
    B, H, Sq, Skv, D = 2, 4, 512, 320, 128
    qkv = torch.randn(B, Sq, 3, H, D, device="cuda", dtype=torch.bfloat16)  # packed projection layout
    q = qkv[:, :, 0].permute(0, 2, 1, 3)
    kfull = torch.randn(B, Skv, H, 2 * D, device="cuda", dtype=torch.bfloat16)
    k = kfull[..., :D].permute(0, 2, 1, 3)
    v = kfull[..., D:].permute(0, 2, 1, 3)
    o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, scale=0.05)
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64")
    ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v), scale=0.05)
    check_forward(o.cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), inputs=(bits(q), bits(k), bits(v)), scale=0.05)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_w64_deferred_max_rescale_paths(dt):
    """Scores that keep rising along the key axis force the reference max to move many times (every rise of more
    than 2^6 triggers the O rescale; lazy: a power-of-two rebase whenever a row sum passes 2^30, with fp16 P 2^6);
    scores that fall leave it untouched; both must match the oracle."""
    import umfa_torch
    torch.manual_seed(4)
    B, H, Sq, Skv, D = 1, 2, 256, 1024, 128
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    base = torch.randn(B, H, Skv, D, device="cuda")
    ramp = torch.linspace(0.0, 1.0, Skv, device="cuda").view(1, 1, Skv, 1)
    # a common direction whose weight grows with the key index: q.k grows by ~40 natural-log units over the row
    direction = q.float().mean(dim=2, keepdim=True)
    direction = direction / direction.norm(dim=-1, keepdim=True)
    for sign in (+1.0, -1.0):
        k = (base * 0.3 + sign * ramp * 60.0 * direction * 11.3).to(dt)
        o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
        assert umfa_torch.last_kernel().startswith("fa_fwd16_w64")
        ref, rl = _oracle().sdpa_forward(npy(q), npy(k), npy(v), return_lse=True)
        assert np.isfinite(o.cpu().numpy()).all()
        # hostile on purpose: the reference moves many times per row (lazy: power-of-two rebases off the row sums)
        check_forward(o.cpu().numpy(), ref, dt, umfa_torch.last_kernel(), f"ramp{sign:+.0f}", scale_max=1.5)
        with umfa_torch.options(softmax_reference="deferred", softmax_tau=6):  # the max-chain bodies: O rescale on every 2^6 rise
            o6 = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
            check_forward(o6.cpu().numpy(), ref, dt, umfa_torch.last_kernel(), f"ramp{sign:+.0f}_tau6", scale_max=1.5)
        assert np.abs(lse.cpu().numpy().reshape(rl.shape) - rl).max() < 5e-2


def test_w64_stale_reference_tail_vs_exact_running_max():
    """The shape on which a stale softmax reference shows its tail (profiles/r2/error_anatomy.md: B1 H64 S1024 against the
    oracle): the default (lazy) and deferred regimes stay inside the stale-reference multiples of the format floor, and
    the same kernel with softmax_reference = exact sits AT the floor -- same rows, same inputs."""
    import umfa_torch
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 64, 1024, 128, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
    rows = np.arange(0, 1024, 4)
    inp = (bits(q), bits(k), bits(v))
    ref = _oracle().sdpa_forward_rows(*inp, rows)
    ol = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64")
    ml, rl = check_forward(ol[:, :, rows].cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), "tail_lazy", inputs=inp, rows=rows)
    with umfa_torch.options(softmax_reference="deferred", softmax_tau=6):
        o6 = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
        check_forward(o6[:, :, rows].cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), "tail_tau6", inputs=inp, rows=rows)
    with umfa_torch.options(softmax_reference="exact"):
        o0 = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
        m0, r0 = check_forward(o0[:, :, rows].cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), "tail_exact", inputs=inp, rows=rows)
    assert r0 < rl and m0 < ml  # the exact reference is the better one: measured 1.05e-3 vs 3.5e-3 max, rms 7 ... 10 % lower
    assert not torch.equal(o0, ol)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_w64_lazy_overflow_restarts_the_segment_with_the_max_chain(dt):
    """The lazy mode's escape hatch: scores that jump by far more than 2^100 (fp16 P: 2^15) between two key tiles overflow
    the stale reference (P = inf); the workgroup must notice, re-run the segment with the max chain, and still meet the
    oracle -- for the rows that jump AND for the ordinary rows that share their workgroup."""
    import umfa_torch
    torch.manual_seed(5)
    B, H, Sq, Skv, D = 1, 3, 512, 768, 128
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    # head 1: keys 300 ... 767 carry a huge component along the mean query direction: scores rise by ~400 nats at key 300
    d = q[:, 1].float().mean(dim=1, keepdim=True)
    d = d / d.norm(dim=-1, keepdim=True)
    kk = k.float()
    kk[:, 1, 300:] += 4000.0 * d
    k = kk.to(dt)
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64")
    ref, rl = _oracle().sdpa_forward(npy(q), npy(k), npy(v), return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, dt, umfa_torch.last_kernel(), "lazy_overflow", scale_max=1.5)
    finite = np.isfinite(rl)
    assert np.abs(lse.cpu().numpy().reshape(rl.shape) - rl)[finite].max() < 5e-2 * max(1.0, np.abs(rl[finite]).max() / 50)
    with umfa_torch.options(softmax_reference="exact"):
        oe = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    # rows of head 1 went through the restart; their neighbours in heads 0 and 2 never left the lazy bodies
    assert float((o[:, 1] - oe[:, 1]).abs().max()) <= 8e-3 * float(oe[:, 1].abs().max())
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32))  # bitwise repeatable


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind", ["causal", "window"])
def test_w64_lazy_underflow_restarts_the_segment(kind, dt):
    """The other escape hatch of the lazy mode.  A row whose first tile of a segment holds no visible key (descending causal
    sweep, band edge of a window) starts from the reference 0; if ALL of its scores then sit hundreds of nats below zero every
    P underflows and l = 0.  The kernel must notice (a row that has keys in the segment and no row sum) and re-run the segment
    with the max chain -- softmax is shift-invariant, the oracle's answer is ordinary."""
    import umfa_torch
    torch.manual_seed(9)
    B, H, S, D = 1, 3, 768, 128
    q = torch.randn(B, H, S, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, S, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, S, D, device="cuda", dtype=dt)
    # head 1: every query gets a unit component along d, every key -4000 along d: all scores drop by ~350 nats; head 2: by
    # ~20 nats only -- nothing for bf16 P, but fp16 P against the reference 0 would be all zero: the fp16 threshold (a row
    # with keys must have collected 2^-6) catches that too
    d = torch.zeros(D, device="cuda")
    d[0] = 1.0
    qq, kk = q.float(), k.float()
    qq[:, 1] = qq[:, 1] + 1.0 * d
    qq[:, 1, :, 0] = qq[:, 1, :, 0].abs() + 0.5
    kk[:, 1] = kk[:, 1] - 4000.0 * d
    qq[:, 2] = qq[:, 2] + 1.0 * d
    qq[:, 2, :, 0] = qq[:, 2, :, 0].abs() + 0.5
    kk[:, 2] = kk[:, 2] - 250.0 * d
    q, k = qq.to(dt), kk.to(dt)
    kw = dict(causal=True) if kind == "causal" else dict(window=(100, 60))
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True, **kw)
    kern = umfa_torch.last_kernel()
    assert kern.startswith("fa_fwd16_w64<"), kern
    assert torch.isfinite(o).all(), "rows with a masked first tile lost every P to underflow"
    if kind == "causal":
        ref = _oracle().sdpa_forward(npy(q), npy(k), npy(v), causal=True)
    else:
        from oracle.oracle import MASK_BOOL
        ref = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=np.ascontiguousarray(_band(S, S, (100, 60), False)), mask_type=MASK_BOOL)
    check_forward(o.cpu().numpy(), ref, dt, kern, "w64_lazy_underflow_" + kind, scale_max=1.5)
    assert torch.isfinite(lse).all()
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, **kw))


@pytest.mark.parametrize("shape", [(1, 2, 256, 256), (1, 2, 512, 512), (2, 3, 768, 768), (1, 2, 1024, 448), (1, 1, 256, 1024),
                                   (1, 4, 2048, 2048)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_w64_causal_vs_oracle(shape, dt):
    """causal (top-left aligned, also Sq != Skv): tiles past the diagonal are skipped, diagonal tiles masked"""
    import umfa_torch
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq * 3 + Skv)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    o, lse = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32, return_lse=True)
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64"), umfa_torch.last_kernel()
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), causal=True, return_lse=True)
    assert np.isfinite(o.cpu().numpy()).all()
    check_forward(o.cpu().numpy(), ref, dt, umfa_torch.last_kernel(), inputs=(npy(q), npy(k), npy(v)), causal=True)
    assert np.abs(lse.cpu().numpy().reshape(ref_lse.shape) - ref_lse).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32))
    o16 = umfa_torch.attention_forward(q, k, v, causal=True)
    assert (o16.float() - o).abs().max() <= (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * o.abs().max() * 1.01


def test_w64_causal_flux_shape_rows():
    import umfa_torch
    torch.manual_seed(1)
    B, H, S, D = 1, 24, 4096, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
    assert fam(umfa_torch.last_kernel()) == "fa_fwd16_w64<bf16,128>"
    assert torch.isfinite(o).all()
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32))
    rows = np.array([0, 1, 63, 64, 255, 256, 1000, 2047, 2048, 4095] + list(range(3000, 3064)))
    ref = _oracle().sdpa_forward_rows(bits(q), bits(k), bits(v), rows, causal=True)
    check_forward(o[:, :, rows].cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), "flux_causal_rows",
                  inputs=(bits(q), bits(k), bits(v)), rows=rows, causal=True)


@pytest.mark.parametrize("shape", [(1, 2, 256, 100, False), (1, 2, 512, 1000, False), (2, 2, 1100, 777, False), (1, 3, 1280, 1100, True),
                                   (1, 2, 1088, 1088, True), (1, 1, 256, 65, True)])
def test_w64_ragged_shapes(shape):
    """Sq not a multiple of 256 (rows past the end are computed on zeros and never stored) and Skv not a multiple of 64
    (the partial last key tile runs the masking variant), with and without the causal mask"""
    import umfa_torch
    B, H, Sq, Skv, causal = shape
    torch.manual_seed(Sq + 7 * Skv)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    # the output tensor sits in the middle of a poisoned allocation: rows past Sq of the last 256-row block, which the
    # kernel computes but must not store, would land in the guard
    pool = torch.full((B * H * Sq * 128 + 2 * 65536,), 7.0, device="cuda", dtype=torch.float32)
    out = pool[65536:65536 + B * H * Sq * 128].view(B, H, Sq, 128)
    o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True, out=out)
    assert bool((pool[:65536] == 7.0).all()) and bool((pool[65536 + B * H * Sq * 128:] == 7.0).all())
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64"), umfa_torch.last_kernel()
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), causal=causal, return_lse=True)
    assert np.isfinite(o.cpu().numpy()).all()
    check_forward(o.cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), inputs=(npy(q), npy(k), npy(v)), causal=causal)
    assert np.abs(lse.cpu().numpy().reshape(ref_lse.shape) - ref_lse).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32))


@pytest.mark.parametrize("shape", [(1, 2, 256, 64, False), (1, 2, 256, 192, False), (2, 3, 512, 256, False), (1, 5, 768, 448, False),
                                   (1, 2, 256, 1024, False), (1, 2, 512, 512, True), (2, 3, 768, 768, True), (1, 2, 1024, 448, True),
                                   (1, 4, 2048, 2048, True), (1, 2, 512, 1000, False), (2, 2, 1100, 777, False),
                                   (1, 3, 1280, 1100, True), (1, 1, 256, 65, True)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_w64_head_dim_64_vs_oracle(shape, dt):
    """head_dim 64 on the same structure (fa_fwd16_w64d64_*: 32 MFMAs per key tile, 128-byte rows in the tile images):
    whole and cut items, causal, ragged Sq / Skv, LSE, both output types, bitwise repeatable"""
    import umfa_torch
    B, H, Sq, Skv, causal = shape
    torch.manual_seed(Sq + 5 * Skv + 64)
    q = torch.randn(B, H, Sq, 64, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, 64, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 64, device="cuda", dtype=dt)
    pool = torch.full((B * H * Sq * 64 + 2 * 65536,), 7.0, device="cuda", dtype=torch.float32)
    out = pool[65536:65536 + B * H * Sq * 64].view(B, H, Sq, 64)
    o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True, out=out)
    kern = umfa_torch.last_kernel()
    assert fam(kern) == ("fa_fwd16_w64<bf16,64>" if dt == torch.bfloat16 else "fa_fwd16_w64<fp16,64>"), kern
    assert bool((pool[:65536] == 7.0).all()) and bool((pool[65536 + B * H * Sq * 64:] == 7.0).all())
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), causal=causal, return_lse=True)
    assert np.isfinite(o.cpu().numpy()).all()
    check_forward(o.cpu().numpy(), ref, dt, kern, "w64_d64", inputs=(npy(q), npy(k), npy(v)), causal=causal)
    assert np.abs(lse.cpu().numpy().reshape(ref_lse.shape) - ref_lse).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32))
    o16 = umfa_torch.attention_forward(q, k, v, causal=causal)
    assert o16.dtype == dt
    assert (o16.float() - o).abs().max() <= (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * o.abs().max() * 1.01
    # the same call on the 128-row kernel (exact running max): two independent kernels, one answer
    with umfa_torch.options(no_w64=1):
        o128 = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<")
    assert float((o - o128).abs().max()) <= (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * float(o128.abs().max())


@pytest.mark.parametrize("mode", ["exact", "deferred", "lazy"])
def test_w64_head_dim_64_softmax_references_and_strides(mode):
    """every softmax-reference policy of the head_dim 64 kernel, on strided [B, S, H, D] storage, 8 key tiles per item"""
    import umfa_torch
    torch.manual_seed(11)
    B, H, S, D = 2, 4, 512, 64
    qs, ks, vs = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) * 2.0 for _ in range(3))
    q, k, v = (t.transpose(1, 2) for t in (qs, ks, vs))
    with umfa_torch.options(softmax_reference=mode):
        o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
        kern = umfa_torch.last_kernel()
        assert fam(kern) == "fa_fwd16_w64<bf16,64>"
        ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v))
        # inputs x 2: peaked rows.  An ideal kernel's dominant P is exactly 1.0 there, a stale reference rounds it like any other (half an ulp of
        # the output on such rows), so the floor-relative bound is asserted for the exact reference only; the format ceiling and 1e-3 hold for all.
        # (Until the launch ran as whole items -- round 4's grid rule -- its 16 items were cut into 128 one-tile parts, each with an exact
        # reference of its own, and the stale modes never showed.)
        check_forward(o.cpu().numpy(), ref, torch.bfloat16, kern, "w64_d64_" + mode, inputs=(bits(q), bits(k), bits(v)) if mode == "exact" else None)


def test_w64_head_dim_64_lazy_overflow_restart():
    import umfa_torch
    torch.manual_seed(6)
    B, H, Sq, Skv, D = 1, 3, 512, 768, 64
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    d = q[:, 1].float().mean(dim=1, keepdim=True)
    d = d / d.norm(dim=-1, keepdim=True)
    kk = k.float()
    kk[:, 1, 300:] += 4000.0 * d
    k = kk.to(torch.bfloat16)
    o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert fam(umfa_torch.last_kernel()) == "fa_fwd16_w64<bf16,64>"
    ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v))
    assert torch.isfinite(o).all()
    check_forward(o.cpu().numpy(), ref, torch.bfloat16, umfa_torch.last_kernel(), "w64_d64_lazy_overflow", scale_max=1.5)
    with umfa_torch.options(softmax_reference="exact"):
        oe = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert float((o[:, 1] - oe[:, 1]).abs().max()) <= 8e-3 * float(oe[:, 1].abs().max())


@pytest.mark.parametrize("shape,window,causal", [((1, 2, 512, 512), (100, 100), False), ((1, 3, 768, 1024), (64, 0), True),
                                                 ((2, 2, 1024, 1024), (256, 0), True), ((1, 2, 1100, 777), (300, 50), False),
                                                 ((1, 2, 512, 512), (0, 0), False), ((1, 1, 512, 256), (10, 10), False),
                                                 ((1, 2, 2048, 2048), (1000, 1000), False), ((1, 2, 1024, 1024), (1024 + 128, 128), False),
                                                 ((1, 5, 1280, 1280), (31, 97), False), ((1, 2, 256, 4096), (500, 700), False),
                                                 ((2, 3, 768, 640), (700, 0), False), ((1, 2, 1024, 512), (900, 30), False)])  # Sq > Skv, left > Skv
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_w64_sliding_window_vs_oracle(shape, window, causal, dt):
    """window=(left, right) on the one-wave-per-SIMD structure (fa_fwd16_w64<.,128,window>): every 256-row block sweeps only
    the key tiles of its band; band-edge tiles (per wave) run the two-compare masking variant; waves whose rows see nothing in
    a tile, rows that see no key at all (O = 0, LSE = -inf), cut items (stream-K over the band's steps), ragged Sq / Skv,
    causal as right = 0"""
    import umfa_torch
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + 3 * Skv + window[0])
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, window=window, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert fam(kern) == ("fa_fwd16_w64<bf16,128,window>" if dt == torch.bfloat16 else "fa_fwd16_w64<fp16,128,window>"), kern
    keep = _band(Sq, Skv, window, causal)
    from oracle.oracle import MASK_BOOL
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=np.ascontiguousarray(keep), mask_type=MASK_BOOL, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    dead = ~keep.any(axis=1)  # rows that see no key: O = 0, LSE = -inf (the masked-row rule of the reference's kernels)
    if dead.any():
        assert (on[:, :, dead] == 0).all() and np.isneginf(lse.cpu().numpy().reshape(B, H, Sq)[:, :, dead]).all()
    live = ~dead
    check_forward(on[:, :, live], ref[:, :, live], dt, kern, "w64_window")
    ln = lse.cpu().numpy().reshape(ref_lse.shape)
    assert np.abs(ln[:, :, live] - ref_lse[:, :, live]).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=causal, window=window, out_dtype=torch.float32))  # bitwise repeatable
    o16 = umfa_torch.attention_forward(q, k, v, causal=causal, window=window)
    assert o16.dtype == dt
    assert (o16.float() - o).abs().max() <= (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * o.abs().max() * 1.01
    # the 128-row kernel's window path (exact running max, bit-identical to the bool band mask): two kernels, one answer
    with umfa_torch.options(no_w64=1):
        o128 = umfa_torch.attention_forward(q, k, v, causal=causal, window=window, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<")
    assert float((o - o128).abs().max()) <= (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * float(o128.abs().max())


@pytest.mark.parametrize("shape,window,causal", [((1, 2, 512, 512), (100, 100), False), ((2, 3, 768, 1024), (64, 0), True),
                                                 ((1, 2, 1100, 777), (300, 50), False), ((2, 2, 768, 640), (700, 0), False)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_w64_sliding_window_head_dim_64(shape, window, causal, dt):
    """the window instantiations of the head_dim 64 family (the same kernel text, W64_DP = 64)"""
    import umfa_torch
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv + 64)
    q = torch.randn(B, H, Sq, 64, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, 64, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 64, device="cuda", dtype=dt)
    o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, window=window, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert fam(kern) == ("fa_fwd16_w64<bf16,64,window>" if dt == torch.bfloat16 else "fa_fwd16_w64<fp16,64,window>"), kern
    keep = _band(Sq, Skv, window, causal)
    from oracle.oracle import MASK_BOOL
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=np.ascontiguousarray(keep), mask_type=MASK_BOOL, return_lse=True)
    live = keep.any(axis=1)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on[:, :, live], ref[:, :, live], dt, kern, "w64_window_d64")
    assert np.abs(lse.cpu().numpy().reshape(ref_lse.shape)[:, :, live] - ref_lse[:, :, live]).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=causal, window=window, out_dtype=torch.float32))


@pytest.mark.parametrize("mode", ["exact", "deferred", "lazy"])
def test_w64_sliding_window_softmax_references(mode):
    """every softmax-reference policy through the window kernel, scores scaled up (|s| of a few nats per sigma): waves whose
    first tiles are fully masked start from the reference 0"""
    import umfa_torch
    torch.manual_seed(21)
    B, H, S = 1, 4, 1536
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", dtype=torch.bfloat16) * 1.7 for _ in range(3))
    win = (200, 120)
    with umfa_torch.options(softmax_reference=mode):
        o = umfa_torch.attention_forward(q, k, v, window=win, out_dtype=torch.float32)
        kern = umfa_torch.last_kernel()
        assert fam(kern) == "fa_fwd16_w64<bf16,128,window>"
        from oracle.oracle import MASK_BOOL
        ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v), mask=np.ascontiguousarray(_band(S, S, win, False)), mask_type=MASK_BOOL)
        check_forward(o.cpu().numpy(), ref, torch.bfloat16, kern, "w64_window_" + mode)


def test_w64_sliding_window_flux_shape_dispatch_and_rows():
    """FLUX shape +-512 without forcing: the dispatcher takes the window kernel; sampled rows against a dense band computation"""
    import umfa_torch
    umfa_torch.set_option("force_w64", 0)
    torch.manual_seed(3)
    B, H, S, D, W = 1, 24, 4096, 128, 512
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out = umfa_torch.attention_forward(q, k, v, window=(W, W), out_dtype=torch.float32)
    kern = umfa_torch.last_kernel()
    assert fam(kern) == "fa_fwd16_w64<bf16,128,window>", kern
    assert torch.equal(out, umfa_torch.attention_forward(q, k, v, window=(W, W), out_dtype=torch.float32))
    for r0 in (0, 448, 2000, 4096 - 64):
        lo, hi_ = max(0, r0 - W), min(S, r0 + 64 + W)
        s = torch.matmul(q[:, :, r0:r0 + 64].double(), k[:, :, lo:hi_].double().transpose(-1, -2)) * D ** -0.5
        i = torch.arange(r0, r0 + 64, device="cuda")[:, None]
        j = torch.arange(lo, hi_, device="cuda")[None, :]
        s = s.masked_fill(~((j <= i + W) & (j >= i - W)), float("-inf"))
        ref = torch.matmul(torch.softmax(s, -1), v[:, :, lo:hi_].double())
        check_forward(out[:, :, r0:r0 + 64].cpu().numpy(), ref.cpu().numpy(), torch.bfloat16, kern, f"w64_window_flux_rows{r0}")


def test_w64_small_ragged_sq_stays_on_the_128_row_kernel():
    import umfa_torch
    q, k, v = (torch.randn(1, 2, 300, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    umfa_torch.attention_forward(q, k, v)
    assert fam(umfa_torch.last_kernel()) == "fa_fwd16<bf16,128>"


@pytest.mark.parametrize("shape,causal,expect_w64", [((1, 24, 4096, 4096), False, True), ((1, 24, 1024, 1024), False, False),
                                                     ((4, 16, 1024, 1024), True, False), ((8, 16, 1024, 1024), True, True),
                                                     ((1, 4, 4096, 4096), False, True), ((1, 16, 2048, 2048), True, False),
                                                     ((1, 256, 256, 256), False, False), ((1, 8, 2048, 2048), False, False), ((1, 16, 2048, 2048), False, True),
                                                     ((2, 32, 1024, 1024), False, True), ((8, 32, 512, 4096), False, False),
                                                     ((2, 24, 1100, 1100), False, False), ((4, 32, 1280, 1280), True, False), ((2, 24, 2100, 2100), False, True)])
def test_w64_dispatch_gate(shape, causal, expect_w64):
    """without UMFA_FORCE_W64: one-workgroup-per-CU kernel only when there is work for (most of) the CUs -- and, for bf16 operands (whose fp16 P V
    product needs the V cast pass), only from four 256-row q-blocks per head on (round 4: with fewer, the pass costs more than the 128-row
    kernel's in-place conversion; profiles/r4/small_nqb_probe.jsonl)"""
    import umfa_torch
    B, H, Sq, Skv = shape
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    umfa_torch.attention_forward(q, k, k, causal=causal)
    torch.cuda.synchronize()
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64") == expect_w64, umfa_torch.last_kernel()


def test_w64_dispatch_gate_fp16_operands_keep_short_query_ranges():
    """fp16 operands need no cast pass: the one-workgroup-per-CU kernel keeps its round-3 gate (whole rounds win at any size)"""
    import umfa_torch
    q = torch.randn(1, 256, 256, 128, device="cuda", dtype=torch.float16)
    k = torch.randn(1, 256, 512, 128, device="cuda", dtype=torch.float16)
    umfa_torch.attention_forward(q, k, k)
    assert umfa_torch.last_kernel() == "fa_fwd16_w64<fp16,128>", umfa_torch.last_kernel()
    umfa_torch.attention_forward(q, q, q)  # (key ranges of fewer than eight tiles: an item is mostly prologue and output)
    assert umfa_torch.last_kernel() == "fa_fwd16<fp16,128>", umfa_torch.last_kernel()


@pytest.mark.parametrize("shape,causal,expect_w64", [((1, 24, 4096, 4096), False, True), ((8, 16, 2048, 2048), False, True), ((8, 16, 1024, 1024), False, False),
                                                     ((1, 40, 1024, 1024), False, False), ((4, 16, 1024, 1024), True, False),
                                                     ((5, 16, 1024, 1024), True, False), ((8, 16, 1024, 1024), True, False), ((8, 16, 2048, 2048), True, False),  # (round 6 refit: 123 us against 225 forced, tools/lab/archive/dbg_mask_d80.py)
                                                     ((4, 16, 4096, 4096), True, True)])
def test_w64_dispatch_gate_head_dim_64(shape, causal, expect_w64):
    """head_dim 64: half the MFMA time per tile step, so the gate sits higher (BASELINE config 2 stays on the 128-row kernel)"""
    import umfa_torch
    B, H, Sq, Skv = shape
    q = torch.randn(B, H, Sq, 64, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 64, device="cuda", dtype=torch.bfloat16)
    umfa_torch.attention_forward(q, k, k, causal=causal)
    torch.cuda.synchronize()
    assert fam(umfa_torch.last_kernel()) == ("fa_fwd16_w64<bf16,64>" if expect_w64 else "fa_fwd16<bf16,64>"), umfa_torch.last_kernel()


@pytest.mark.parametrize("causal", [False, True])
def test_w64_very_long_sequence_rows(causal):
    """S = 131072 (B1 H2 D128: 2048 key tiles per item, 33 MB slabs, 32-bit offsets well past 2^24): sampled rows against an
    fp64 restatement on the GPU; LSE; bitwise repeatable"""
    import umfa_torch
    umfa_torch.set_option("force_w64", 0)
    torch.manual_seed(31)
    B, H, S, D = 1, 2, 131072, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert fam(kern) == "fa_fwd16_w64<bf16,128>", kern
    assert torch.isfinite(out).all()
    for r0 in (0, 255, 65536 - 32, 100000, S - 64):
        rows = slice(r0, r0 + 64)
        s = torch.matmul(q[:, :, rows].double(), k.double().transpose(-1, -2)) * D ** -0.5
        if causal:
            i = torch.arange(r0, r0 + 64, device="cuda")[:, None]
            j = torch.arange(S, device="cuda")[None, :]
            s = s.masked_fill(j > i, float("-inf"))
        ref = torch.matmul(torch.softmax(s, -1), v.double())
        check_forward(out[:, :, rows].cpu().numpy(), ref.cpu().numpy(), torch.bfloat16, kern, f"w64_S131072_rows{r0}")
        assert (lse.view(B, H, S)[:, :, rows].double() - torch.logsumexp(s, -1)).abs().max().item() < 2e-2
    del s, ref
    assert torch.equal(out, umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32))


@pytest.mark.parametrize("shape,causal", [((1, 2, 256, 256), False), ((2, 3, 768, 448), False), ((1, 2, 1100, 777), False),
                                          ((1, 3, 1024, 1024), True), ((1, 2, 512, 1000), True)])
def test_w64_bf16_operands_with_fp16_pv(shape, causal):
    """the default bf16 forward (option pv_fp16, on): bf16 Q / K / V, P rounded to fp16 and multiplied with V converted bf16 -> fp16
    inside the kernel (exact over fp16's range): the second product carries 11 bits instead of 8 and the bf16-input forward sits
    INSIDE the north-star's 1e-3 -- held here to fp16's own format ceiling (one ulp of P at 1.0 = 2^-11) against the oracle on the
    bf16 inputs"""
    import umfa_torch
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    vs = torch.randn(B, Skv, H, 128, device="cuda", dtype=torch.bfloat16)
    v = vs.transpose(1, 2)  # strided V: the kernel's V loads take the caller's strides
    ref, ref_lse = _oracle().sdpa_forward(bits(q), bits(k), bits(v), causal=causal, return_lse=True)
    o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern == "fa_fwd16_w64<bf16,128,pv16>", kern
    o16 = umfa_torch.attention_forward(q, k, v, causal=causal)
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32))
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    mx = float(np.abs(on - ref).max() / np.abs(ref).max())
    assert mx < 2.0 ** -11, mx  # measured 2-3e-4; the bf16 P V kernel: 1-2e-3
    assert np.abs(lse.cpu().numpy().reshape(ref_lse.shape) - ref_lse).max() < 2e-2
    assert o16.dtype == torch.bfloat16 and (o16.float() - o).abs().max() <= 2.0 ** -8 * o.abs().max() * 1.01
    with umfa_torch.options(pv_fp16=0):
        o_plain = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32)
        assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128>"
    assert float(np.abs(o_plain.cpu().numpy() - ref).max() / np.abs(ref).max()) > mx  # what the fp16 product buys


@pytest.mark.parametrize("causal", [False, True])
def test_w64_fp16_pv_head_dim_64(causal):
    import umfa_torch
    torch.manual_seed(64)
    q, k, v = (torch.randn(2, 3, 768, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    ref = _oracle().sdpa_forward(bits(q), bits(k), bits(v), causal=causal)
    o = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32)
    assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,64,pv16>"
    assert float(np.abs(o.cpu().numpy() - ref).max() / np.abs(ref).max()) < 2.0 ** -11


def test_w64_fp16_pv_flux_rows_and_range():
    """the default bf16 forward at the FLUX shape inside 1e-3 (with a factor of two to spare) -- and for EVERY bf16 V: the cast pre-pass
    shifts each (batch, head) slab by its own power of two, so a 3e8 outlier (round 4: non-finite outputs + a sticky context-wide
    fall-back) and a V of 1e-6 (round 4: a coarse image) come out finite and inside the tolerance, on the in-stream entry, with
    nothing for the host to read (tests/test_gpu_pv16_range.py: the same through a replayed hipGraph and on the 128-row kernel)"""
    import umfa_torch
    umfa_torch.set_option("force_w64", 0)
    torch.manual_seed(2)
    B, H, S, D = 1, 24, 4096, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    from oracle import parity
    rows = parity.sample_rows(S)
    ref = _oracle().sdpa_forward_rows(bits(q), bits(k), bits(v), rows)
    o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128,pv16>"
    mx = float(np.abs(o[:, :, rows].cpu().numpy() - ref).max() / np.abs(ref).max())
    assert mx < 1.0e-3 / 2, mx  # the north-star's bound with a factor of two to spare
    # per-slab shifts: head 0 holds a 3e8 outlier, head 1 is ~1e-6, head 2 ~1e4, head 3 mixes 1e-20 rows into ordinary ones
    vx = v.clone()
    vx[0, 0, 5, 7] = 3.0e8
    vx[0, 1] = (v[0, 1].float() * 1e-6).to(torch.bfloat16)
    vx[0, 2] = (v[0, 2].float() * 1e4).to(torch.bfloat16)
    vx[0, 3, ::3] = (v[0, 3, ::3].float() * 1e-20).to(torch.bfloat16)
    ox = umfa_torch.attention_forward(q, k, vx, out_dtype=torch.float32)
    assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128,pv16>"
    assert torch.isfinite(ox).all()
    refx = _oracle().sdpa_forward_rows(bits(q[:, :4]), bits(k[:, :4]), bits(vx[:, :4]), rows)
    oxn = ox[:, :4, rows].cpu().numpy()
    for h in range(4):  # every slab against ITS OWN scale
        e = float(np.abs(oxn[0, h] - refx[0, h]).max() / np.abs(refx[0, h]).max())
        assert e < 1.0e-3 / 2, (h, e)
    # the outlier's head, the columns the outlier does not touch: ordinary values sit 2^28 below the slab's largest there -- still exact in fp16
    cols = [c for c in range(D) if c != 7]
    e = float(np.abs(oxn[0, 0][:, cols] - refx[0, 0][:, cols]).max() / np.abs(refx[0, 0][:, cols]).max())
    assert e < 1.0e-3, e
    assert torch.equal(ox[:, 4:], o[:, 4:])  # the other heads: the same bits as before (their slabs' shift did not move)
    # the same call as two launches (amax, then cast): what slabs of more than CUs / 2 chunks take
    with umfa_torch.options(cast_two_pass=1):
        assert torch.equal(umfa_torch.attention_forward(q, k, vx, out_dtype=torch.float32), ox)


