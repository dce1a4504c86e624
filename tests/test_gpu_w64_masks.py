"""Bool mask tensors on the one-wave-per-SIMD forward (fa_fwd16_w64<., 128, mask>, round 4): the runtime re-packs the caller's mask
(any <= 4-D broadcastable bool tensor with its own strides) into per-lane bit words, per-wave tile classes and the visited-tile list
of every 256-row block (fa_aux.hip mask_pack_kernel / mask_list_kernel); the kernel sweeps the list.  Semantics are the
reference's (MFABridge.swift:157-242: non-zero attends, size-1 dims broadcast, a row with every key masked gives O = 0,
LSE = -inf), checked against the CPU oracle WITH the mask, against the 128-row kernel (option no_w64_mask) and for run-to-run
bitwise repeatability."""
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


def _mask(kind, B, H, Sq, Skv, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    i = torch.arange(Sq, device="cuda")[:, None]
    j = torch.arange(Skv, device="cuda")[None, :]
    if kind == "random_per_head":
        m = torch.rand(B, H, Sq, Skv, device="cuda", generator=g) < 0.7
        m[..., 0] = True
        return m
    if kind == "random_2d":
        m = torch.rand(Sq, Skv, device="cuda", generator=g) < 0.5
        m[:, 3] = True
        return m
    if kind == "padding":                      # [B, 1, 1, Skv]: per-batch key padding
        lens = torch.tensor([max(1, Skv - 37 - 211 * b) for b in range(B)], device="cuda")
        return (j[None] < lens[:, None, None])[:, None]
    if kind == "blockdiag":                    # [1, 1, Sq, Skv]: documents of 192 rows / 160 keys (not on the tile grid)
        return ((i // 192) == (j // 160))[None, None]
    if kind == "one_tile":                     # [1, 1, 1, Skv]: every block lists ONE key tile (fewer shared steps than workgroups)
        return (j < 10)[None, None]
    if kind == "all_open":
        return torch.ones(1, 1, Sq, Skv, dtype=torch.bool, device="cuda")
    if kind == "empty_rows_and_blocks":        # rows 5.., a whole 256-row block and a whole head see nothing
        m = torch.rand(B, H, Sq, Skv, device="cuda", generator=g) < 0.6
        m[:, :, 5::17] = False
        m[:, 0, 256:512] = False
        if H > 1:
            m[:, 1] = False
        return m
    if kind == "strided_view":                 # a non-contiguous mask: every second column of a wider tensor
        wide = torch.rand(B, 1, Sq, 2 * Skv, device="cuda", generator=g) < 0.6
        wide[..., 0] = True
        return wide[..., ::2]
    raise ValueError(kind)


KINDS = ["random_per_head", "random_2d", "padding", "blockdiag", "all_open", "empty_rows_and_blocks", "strided_view"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 2, 512, 512), (1, 3, 1280, 777)])
@pytest.mark.parametrize("D", [128, 64])  # (head_dim 64: round 5)
def test_w64_mask_tensor_vs_oracle(kind, dt, shape, D, umfa_opts):
    import umfa_torch
    umfa_opts(force_w64=1)
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    m = _mask(kind, B, H, Sq, Skv, seed=Sq)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern in (f"fa_fwd16_w64<bf16,{D},pv16,mask>", f"fa_fwd16_w64<fp16,{D},mask>"), kern
    mfull = m.expand(B, H, Sq, Skv) if m.dim() == 4 else m.expand(Sq, Skv)
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=np.ascontiguousarray(mfull.cpu().numpy()),
                                          mask_type=_oracle().MASK_BOOL, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, dt, kern, f"w64_mask_{kind}")
    dead = ~mfull.expand(B, H, Sq, Skv).any(-1).cpu().numpy() if m.dim() == 4 else np.broadcast_to(~mfull.any(-1).cpu().numpy(), (B, H, Sq))
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert (on[dead] == 0).all() and np.isneginf(ln[dead]).all()          # rows that see no key: O = 0, LSE = -inf
    assert np.abs(ln[~dead] - ref_lse[~dead]).max() < 2e-2
    # bitwise repeatable, and the same numbers class as the 128-row kernel's tile-flag path
    o2 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
    assert torch.equal(o, o2)
    with umfa_torch.options(no_w64_mask=1, force_w64=0):
        o3 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    assert float((o - o3).abs().max()) <= 2.0 ** -9 * float(o3.abs().max())
    # 16-bit output epilogue
    o16 = umfa_torch.attention_forward(q, k, v, mask=m)
    assert o16.dtype == dt and float((o16.float() - o).abs().max()) <= 2.0 ** -8 * float(o.abs().max()) * 1.01


@pytest.mark.parametrize("kind", ["random_per_head", "random_2d", "padding", "blockdiag", "all_open", "empty_rows_and_blocks", "strided_view"])
@pytest.mark.parametrize("dt,D", [(torch.bfloat16, 128), (torch.float16, 64)])
@pytest.mark.parametrize("shape,grid", [((2, 2, 512, 512), 0), ((1, 3, 1280, 777), 0), ((1, 3, 1280, 1400), 5)])
def test_w64_mask_with_the_causal_flag(kind, dt, D, shape, grid, umfa_opts):
    """causal AND a bool mask tensor (a causal LM with key padding, packed documents): the pre-pass folds key <= row into the bits it packs, so
    the mask kernel -- which has no causal instantiation -- sweeps lists that never contain a tile above the diagonal.  Oracle with both,
    rows that see nothing (O = 0, LSE = -inf), cut blocks, and the 128-row kernel's numbers class."""
    import umfa_torch
    umfa_opts(force_w64=1)
    if grid:
        umfa_opts(w64_grid=grid)
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv + D)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt)
    m = _mask(kind, B, H, Sq, Skv, seed=Sq + 1)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, causal=True, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern in (f"fa_fwd16_w64<bf16,{D},pv16,mask>", f"fa_fwd16_w64<fp16,{D},mask>"), kern
    mfull = (m.expand(B, H, Sq, Skv) if m.dim() == 4 else m.expand(Sq, Skv)[None, None].expand(B, H, Sq, Skv)).clone()
    mfull &= (torch.arange(Skv, device="cuda")[None, :] <= torch.arange(Sq, device="cuda")[:, None])
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=np.ascontiguousarray(mfull.cpu().numpy()),
                                          mask_type=_oracle().MASK_BOOL, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, dt, kern, f"w64_mask_causal_{kind}")
    dead = ~mfull.any(-1).cpu().numpy()
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert (on[dead] == 0).all() and np.isneginf(ln[dead]).all()
    assert np.abs(ln[~dead] - ref_lse[~dead]).max() < 2e-2
    o2 = umfa_torch.attention_forward(q, k, v, mask=m, causal=True, out_dtype=torch.float32)
    assert torch.equal(o, o2)
    with umfa_torch.options(no_w64_mask=1, force_w64=0, w64_grid=0):
        o3 = umfa_torch.attention_forward(q, k, v, mask=m, causal=True, out_dtype=torch.float32)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    assert float((o - o3).abs().max()) <= 2.0 ** -9 * float(o3.abs().max())


@pytest.mark.parametrize("kind", ["random_per_head", "padding", "blockdiag", "empty_rows_and_blocks", "strided_view", "one_tile"])
@pytest.mark.parametrize("shape,grid", [((2, 2, 512, 512), 3), ((2, 2, 512, 512), 5), ((1, 3, 1280, 777), 4), ((1, 3, 1280, 777), 7), ((1, 3, 1280, 777), 14)])
@pytest.mark.parametrize("D", [128, 64])
def test_w64_mask_cut_blocks(kind, shape, grid, D, umfa_opts):
    """more blocks than workgroups with a remainder: the last n % grid blocks are cut ALONG THEIR TILE LISTS into grid equal slices (every
    workgroup scans the running sums of their list lengths itself) and folded like the unmasked kernel's cut items.  Forced here with
    the lab option w64_grid on small shapes; the FLUX-shape test below meets it by itself (384 blocks on 256 CUs)."""
    import umfa_torch
    B, H, Sq, Skv = shape
    torch.manual_seed(Sq + Skv + grid)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    m = _mask(kind, B, H, Sq, Skv, seed=Sq + grid)
    umfa_opts(force_w64=1, w64_grid=grid)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern == f"fa_fwd16_w64<bf16,{D},pv16,mask>", kern
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32))  # whoever folds: the same bits
    mfull = m.expand(B, H, Sq, Skv)
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=np.ascontiguousarray(mfull.cpu().numpy()),
                                          mask_type=_oracle().MASK_BOOL, return_lse=True)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    check_forward(on, ref, torch.bfloat16, kern, f"w64_mask_cut_{kind}")
    dead = ~mfull.any(-1).cpu().numpy()
    ln = lse.cpu().numpy().reshape(B, H, Sq)
    assert (on[dead] == 0).all() and np.isneginf(ln[dead]).all()
    assert np.abs(ln[~dead] - ref_lse[~dead]).max() < 2e-2
    # whole blocks only (grid = number of blocks): a part rounds its P against its own running reference, so the two results differ like
    # two fp16-P kernels do -- each within half an ulp of P at 1.0 of the truth
    umfa_opts(force_w64=1, w64_grid=0)
    o_whole = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
    assert float((o - o_whole).abs().max()) <= 2.0 ** -10 * float(o_whole.abs().max())


@pytest.mark.parametrize("kind", ["padding", "blockdiag"])
def test_w64_mask_flux_shape_rows(kind, umfa_opts):
    """the bench's masked FLUX entries (B1 H24 S4096 D128): default routing takes the mask kernel (384 items >= the CU count),
    rows against the oracle with the mask, inside the north-star's 1e-3"""
    import umfa_torch
    umfa_opts(force_w64=0)
    from oracle import parity
    B, H, S, D = 1, 24, 4096, 128
    torch.manual_seed(5)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    m = (i < 3000)[None, None, None, :].contiguous() if kind == "padding" else ((i[:, None] // 1024) == (i[None, :] // 1024))[None, None].contiguous()
    o = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
    assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128,pv16,mask>", umfa_torch.last_kernel()
    rows = parity.sample_rows(S)
    mrows = np.ascontiguousarray(np.broadcast_to(m[0, 0].cpu().numpy()[rows if m.shape[2] > 1 else [0] * len(rows)], (len(rows), S)))
    ref = _oracle().sdpa_forward(np.ascontiguousarray(bits(q)[:, :, rows]), bits(k), bits(v), mask=mrows, mask_type=_oracle().MASK_BOOL)
    err = float(np.abs(o[:, :, rows].cpu().numpy() - ref).max() / np.abs(ref).max())
    assert err < 1.0e-3, err
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32))


@pytest.mark.parametrize("seed", range(24))
def test_w64_mask_schedule_fuzz(seed, umfa_opts):
    """random shapes x random bool masks (any broadcast pattern, any density, empty rows / blocks / heads) x a random number of workgroups:
    whole rounds, cut blocks, more workgroups than shared steps, one workgroup for everything.  Against the 128-row kernel's tile-flag
    path on the same inputs (two fp16-P kernels: each within half an ulp of P at 1.0 of the truth) and for bitwise repeatability."""
    import umfa_torch
    rng = np.random.default_rng(1000 + seed)
    B, H = int(rng.integers(1, 3)), int(rng.integers(1, 5))
    Sq = int(rng.choice([256, 512, 768, 1024, 1280, 1100, 1536]))
    Skv = int(rng.choice([64, 100, 256, 333, 512, 777, 1024, 1500]))
    dt = torch.bfloat16 if seed % 3 else torch.float16
    torch.manual_seed(seed)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=dt)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    mb, mh, mq = (B if rng.random() < 0.5 else 1), (H if rng.random() < 0.5 else 1), (Sq if rng.random() < 0.7 else 1)
    dens = float(rng.choice([0.02, 0.3, 0.7, 0.97]))
    g = torch.Generator(device="cuda").manual_seed(seed)
    m = torch.rand(mb, mh, mq, Skv, device="cuda", generator=g) < dens
    if rng.random() < 0.5:  # structure on top: a band of fully masked key tiles, a fully masked row range
        lo = int(rng.integers(0, Skv))
        m[..., lo:lo + int(rng.integers(1, 300))] = False
        if mq > 1:
            r0 = int(rng.integers(0, Sq))
            m[:, :, r0:r0 + int(rng.integers(1, 400))] = False
    items = B * H * ((Sq + 255) // 256)
    grid = int(rng.integers(1, min(items, 16) + 1))
    umfa_opts(force_w64=1, w64_grid=grid)
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern.endswith(",mask>"), (kern, B, H, Sq, Skv, grid)
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)), (B, H, Sq, Skv, grid)
    assert torch.isfinite(o).all()
    with umfa_torch.options(no_w64_mask=1, force_w64=0, w64_grid=0):
        o2, lse2 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
        assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()
    dead = ~m.expand(B, H, Sq, Skv).any(-1)
    assert (o[dead] == 0).all() and torch.isneginf(lse.reshape(B, H, Sq)[dead]).all()
    scale = float(o2.abs().max()) if float(o2.abs().max()) > 0 else 1.0
    assert float((o - o2).abs().max()) <= 2.0 ** -10 * scale, (B, H, Sq, Skv, grid, dens)
    live = ~dead
    if live.any():
        assert float((lse.reshape(B, H, Sq)[live] - lse2.reshape(B, H, Sq)[live]).abs().max()) < 1e-3


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gain", [1.0, 6.0, 20.0])
def test_w64_mask_key_padding_lazy_reference(dt, gain, umfa_opts):
    """masks without a row dimension run the LAZY tile bodies (no row max after a segment's first tile; rebases and the give-up path are read
    off the row sums): hostile score ranges (gain 20: jumps of hundreds of nats between tiles -> rebases, overflow hand-shake, re-run
    with the max chain) against the oracle with the mask, and against the same launch with the max chain (option no_w64_mask_lazy)"""
    import umfa_torch
    umfa_opts(force_w64=1, w64_grid=3)
    B, H, Sq, Skv = 2, 2, 512, 1100
    torch.manual_seed(int(gain))
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=dt) * gain
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=dt)
    j = torch.arange(Skv, device="cuda")
    lens = torch.tensor([1000, 333], device="cuda")
    m = ((j[None] < lens[:, None]) & (j[None] % 7 != 3))[:, None, None, :]  # [B, 1, 1, Skv]: padding + holes
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern.endswith(",mask>"), kern
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32))
    mfull = m.expand(B, H, Sq, Skv)
    ref, ref_lse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), mask=np.ascontiguousarray(mfull.cpu().numpy()),
                                          mask_type=_oracle().MASK_BOOL, return_lse=True)
    check_forward(o.cpu().numpy(), ref, dt, kern, f"w64_mask_lazy_gain{gain}", scale_max=1.0 if gain == 1.0 else 2.0)
    assert np.abs(lse.cpu().numpy().reshape(B, H, Sq) - ref_lse).max() < 2e-2 * max(1.0, gain)
    with umfa_torch.options(no_w64_mask_lazy=1):
        o2 = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32)
        assert umfa_torch.last_kernel() == kern
    assert float((o - o2).abs().max()) <= 2.0 ** -9 * float(o2.abs().max())


def test_w64_mask_few_blocks_key_padding(umfa_opts):
    """fewer 256-row blocks than CUs (B1 H8 S4096: 128): a mask WITHOUT a row dimension (key padding) takes the mask kernel by default, every
    block shared between workgroups and folded; a [Sq, Skv] mask of the same call stays on the 128-row kernel (how dense it is the host
    cannot know).  Rows against the oracle with the mask, inside 1e-3; bitwise repeatable."""
    import umfa_torch
    umfa_opts(force_w64=0)
    from oracle import parity
    B, H, S, D = 1, 8, 4096, 128
    torch.manual_seed(9)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    m = (i < 3333)[None, None, None, :].contiguous()
    o, lse = umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32, return_lse=True)
    if torch.cuda.get_device_properties(0).multi_processor_count * 10 <= B * H * (S // 256) * (S // 64):
        assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,128,pv16,mask>", umfa_torch.last_kernel()
    rows = parity.sample_rows(S)
    mrows = np.ascontiguousarray(np.broadcast_to(m[0, 0].cpu().numpy(), (len(rows), S)))
    ref, ref_lse = _oracle().sdpa_forward(np.ascontiguousarray(bits(q)[:, :, rows]), bits(k), bits(v), mask=mrows, mask_type=_oracle().MASK_BOOL, return_lse=True)
    err = float(np.abs(o[:, :, rows].cpu().numpy() - ref).max() / np.abs(ref).max())
    assert err < 1.0e-3, err
    assert np.abs(lse.reshape(B, H, S)[:, :, rows].cpu().numpy() - ref_lse).max() < 2e-2
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, mask=m, out_dtype=torch.float32))
    m2 = ((i[:, None] // 1024) == (i[None, :] // 1024))[None, None].contiguous()
    umfa_torch.attention_forward(q, k, v, mask=m2)
    assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()


def test_w64_mask_routing_gate(umfa_opts):
    """few items (fewer 256-row blocks than CUs), additive masks and bf16 P V stay on the 128-row kernel; causal + mask: the mask kernel, the
    causal flag folded into the packed bits (late round 5)"""
    import umfa_torch
    umfa_opts(force_w64=0)
    torch.manual_seed(1)
    q, k, v = (torch.randn(1, 4, 512, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    mb = torch.rand(1, 1, 512, 512, device="cuda") < 0.5
    mb[..., 0] = True
    umfa_torch.attention_forward(q, k, v, mask=mb)
    assert umfa_torch.last_kernel().startswith("fa_fwd16<"), umfa_torch.last_kernel()     # 8 items
    umfa_opts(force_w64=1)
    umfa_torch.attention_forward(q, k, v, mask=mb)
    assert umfa_torch.last_kernel().endswith(",mask>")
    umfa_torch.attention_forward(q, k, v, mask=mb, causal=True)
    assert umfa_torch.last_kernel().endswith(",mask>")
    umfa_torch.attention_forward(q, k, v, mask=torch.zeros(1, 1, 512, 512, device="cuda"))
    # (an fp32 additive mask under force_w64: the guarded pair since the end of round 6 -- tests/test_gpu_w64_f32_mask.py; without force, 8 items stay on the 128-row kernel)
    assert " | fa_fwd16<" in umfa_torch.last_kernel(), umfa_torch.last_kernel()
    with umfa_torch.options(force_w64=0):
        umfa_torch.attention_forward(q, k, v, mask=torch.zeros(1, 1, 512, 512, device="cuda"))
        assert umfa_torch.last_kernel().startswith("fa_fwd16<")
    with umfa_torch.options(pv_fp16=0):
        umfa_torch.attention_forward(q, k, v, mask=mb)
        assert umfa_torch.last_kernel() == "fa_fwd16<bf16,128>"
    q6, k6, v6 = (torch.randn(1, 4, 512, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    umfa_torch.attention_forward(q6, k6, v6, mask=mb)
    assert umfa_torch.last_kernel() == "fa_fwd16_w64<bf16,64,pv16,mask>"  # (head_dim 64 has the mask instantiations since round 5; forced here)
    umfa_opts(force_w64=0)
    umfa_torch.attention_forward(q6, k6, v6, mask=mb)
    assert umfa_torch.last_kernel().startswith("fa_fwd16<")  # 8 blocks: the gate
