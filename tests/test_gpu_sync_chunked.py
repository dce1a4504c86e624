"""The synchronous entry on HOST-wrapping buffers in head chunks (runtime.hip forward_sync_chunked; mfa_attention_forward,
MFABridge.swift:1074-1433): the chunks' uploads, kernels and downloads overlap on side streams.  Same results as the
one-upload form (option sync_chunks = 1) and as the oracle, for every way the chunk plan cuts (head ranges of one batch,
whole batches, remainders), with LSE, with a mask that has no batch / head extent, and causal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    import umfa
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a device: the product path has no CPU fallback")
    c = umfa.MFAContext()
    yield c
    c.close()


def bf16_bits(rng, shape):
    return (rng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def rel(a, ref):
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


def run(ctx, q, k, v, chunks, expect_chunked=None, **kw):
    import umfa
    import umfa_torch
    n0 = int(umfa_torch.get_option("sync_chunked_calls"))
    with umfa_torch.options(sync_chunks=chunks):
        out = umfa.flash_attention_forward(ctx, q, k, v, input_precision="bf16", intermediate_precision="bf16", layout="bhsd", **kw)
    took = int(umfa_torch.get_option("sync_chunked_calls")) - n0
    assert took == int(chunks != 1 if expect_chunked is None else expect_chunked), (took, chunks)  # which form ran
    return out


# (B, H, S, D, chunks): >= 16 MB over the link each (the chunked form's floor), >= 1 MiB per operand (pinned ranges)
PLANS = [
    (1, 24, 1024, 128, 6),   # head ranges of one batch: 6 x 4 heads
    (1, 20, 1024, 128, 6),   # ... with a remainder: 5 x 4 heads
    (2, 10, 1024, 128, 6),   # two batches x 3 head ranges (4, 4, 2)
    (8, 3, 1024, 128, 6),    # B >= chunks: whole batches, 2 at a time
    (7, 3, 1024, 128, 3),    # ... 3 + 3 + 1 batches
    (1, 16, 2048, 64, 16),   # head_dim 64, one head per chunk
    (1, 32, 2048, 128, 0),   # the default: by size (84 MB over the link -> 6 chunks)
]


@pytest.mark.parametrize("B,H,S,D,chunks", PLANS)
def test_chunked_equals_one_upload_and_oracle(ctx, B, H, S, D, chunks):
    from oracle import oracle
    rng = np.random.default_rng(B * 1000 + H * 10 + chunks)
    q, k, v = (bf16_bits(rng, (B, H, S, D)) for _ in range(3))
    assert 3 * q.nbytes + 2 * q.nbytes >= 16 << 20 and q.nbytes >= 1 << 20
    o1 = run(ctx, q, k, v, 1)
    oc = run(ctx, q, k, v, chunks)
    assert oc.shape == q.shape and oc.dtype == np.float32 and np.isfinite(oc).all()
    # (a chunk is a launch of its own over fewer heads: another kernel of the routing table's, or another split of the persistent kernel's items --
    # each inside the north-star's bound of the oracle, below; not bit for bit)
    assert rel(oc, o1) < 1e-3, rel(oc, o1)
    # the oracle on the first and the last (batch, head) slab and one in the middle
    for b, h in ((0, 0), (B - 1, H - 1), (B // 2, H // 2)):
        ref = oracle.sdpa_forward(q[b:b + 1, h:h + 1], k[b:b + 1, h:h + 1], v[b:b + 1, h:h + 1])
        assert rel(oc[b:b + 1, h:h + 1], ref) < 1e-3  # the north-star's tolerance (bf16 operands, P V product in fp16)


def test_chunked_with_lse_and_causal(ctx):
    from oracle import oracle
    rng = np.random.default_rng(7)
    B, H, S, D = 1, 12, 1024, 128
    q, k, v = (bf16_bits(rng, (B, H, S, D)) for _ in range(3))
    q = np.concatenate([q, q], axis=1)[:, :24]
    k = np.concatenate([k, k], axis=1)[:, :24]
    v = np.concatenate([v, v], axis=1)[:, :24]
    q, k, v = (np.ascontiguousarray(x) for x in (q, k, v))
    o1, l1 = run(ctx, q, k, v, 1, causal=True, return_lse=True)
    oc, lc = run(ctx, q, k, v, 6, causal=True, return_lse=True)
    assert rel(oc, o1) < 1e-3 and np.abs(lc - l1).max() < 1e-3
    # heads h and h + 12 hold the same data: a chunk boundary lies between them
    assert rel(oc[:, :12], oc[:, 12:]) < 1e-3
    for h in (0, 23):
        ref, rl = oracle.sdpa_forward(q[:, h:h + 1], k[:, h:h + 1], v[:, h:h + 1], causal=True, return_lse=True)
        assert rel(oc[:, h:h + 1], ref) < 1e-3
        assert np.abs(lc.reshape(B, 24, S)[:, h] - rl.reshape(B, S)).max() < 2e-3


def test_chunked_with_a_shared_mask(ctx):
    from oracle import oracle
    rng = np.random.default_rng(11)
    B, H, S, D = 1, 24, 1024, 128
    q, k, v = (bf16_bits(rng, (B, H, S, D)) for _ in range(3))
    mask = rng.random((S, S)) < 0.7  # True = attend; [Sq, Skv]: no batch / head extent -> the chunked form takes it
    mask[:, 0] = True
    o1 = run(ctx, q, k, v, 1, attn_mask=mask)
    oc = run(ctx, q, k, v, 6, attn_mask=mask)
    assert rel(oc, o1) < 1e-3
    ref = oracle.sdpa_forward(q[:, 5:6], k[:, 5:6], v[:, 5:6], mask=mask[None, None], mask_type=oracle.MASK_BOOL)
    assert rel(oc[:, 5:6], ref) < 1e-3
    # a mask WITH a head extent stays on the one-upload form: same numbers either way
    mh = np.broadcast_to(mask, (1, H, S, S)).copy()
    mh[0, 3] = True
    oh = run(ctx, q, k, v, 6, expect_chunked=False, attn_mask=mh)
    assert rel(np.delete(oh, 3, axis=1), np.delete(o1, 3, axis=1)) < 1e-3
    assert rel(oh[:, 3:4], oracle.sdpa_forward(q[:, 3:4], k[:, 3:4], v[:, 3:4])) < 1e-3


def test_latency_counts_the_chunks_kernels(ctx):
    rng = np.random.default_rng(3)
    q, k, v = (bf16_bits(rng, (1, 24, 1024, 128)) for _ in range(3))
    run(ctx, q, k, v, 6)
    t = ctx.gpu_latency
    assert 0 < t < 5e-3  # seconds of kernels (mfa_get_gpu_latency), not of copies


def test_small_calls_and_transposed_operands_stay_on_one_upload(ctx):
    import umfa
    import umfa_torch
    rng = np.random.default_rng(5)
    q, k, v = (bf16_bits(rng, (1, 4, 256, 64)) for _ in range(3))  # 128 KiB per operand: below the chunked form's floor (and not pinned)
    n0 = int(umfa_torch.get_option("sync_chunked_calls"))
    o = umfa.flash_attention_forward(ctx, q, k, v, input_precision="bf16", intermediate_precision="bf16", layout="bhsd")
    assert int(umfa_torch.get_option("sync_chunked_calls")) == n0 and np.isfinite(o).all()


@pytest.mark.parametrize("B,H,S,D,chunks,causal", [(1, 12, 1024, 128, 6, False), (1, 24, 1024, 128, 0, True), (4, 3, 1024, 64, 4, True), (2, 5, 1024, 128, 6, False)])
def test_backward_chunked_equals_one_upload_and_oracle(ctx, B, H, S, D, chunks, causal):
    """mfa_attention_backward (MFABridge.swift:3171-3282) on host arrays: the same chunk plan, gradients against the one-upload form and the oracle."""
    import umfa
    import umfa_torch
    from oracle import oracle
    rng = np.random.default_rng(100 + B + H + chunks)
    q, k, v, do = (bf16_bits(rng, (B, H, S, D)) for _ in range(4))
    with umfa_torch.options(sync_chunks=1):
        o, lse = umfa.flash_attention_forward(ctx, q, k, v, input_precision="bf16", intermediate_precision="bf16", layout="bhsd", causal=causal, return_lse=True)

    def bwd(c):
        n0 = int(umfa_torch.get_option("sync_chunked_calls"))
        with umfa_torch.options(sync_chunks=c):
            g = umfa.attention_backward(ctx, do, q, k, v, o, lse, causal=causal, input_precision="bf16", layout="bhsd")
        assert int(umfa_torch.get_option("sync_chunked_calls")) - n0 == int(c != 1), c
        assert ctx.last_kernel.startswith("fa_bwd16"), ctx.last_kernel
        return g

    g1, gc = bwd(1), bwd(chunks)
    for a, b_, name in zip(gc, g1, ("dq", "dk", "dv", "D")):
        assert np.isfinite(a).all()
        assert rel(a, b_) < 2e-3, (name, rel(a, b_))  # (the 16-bit engine's own repeatability across launch shapes; BWD16_TOL against the oracle is 8e-3)
    for b, h in ((0, 0), (B - 1, H - 1)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        ro, rl = oracle.sdpa_forward(q[sl], k[sl], v[sl], causal=causal, return_lse=True)
        rdq, rdk, rdv, _ = oracle.sdpa_backward(do[sl], q[sl], k[sl], v[sl], ro, rl, causal=causal)
        for a, r, name in ((gc[0][sl], rdq, "dq"), (gc[1][sl], rdk, "dk"), (gc[2][sl], rdv, "dv")):
            assert rel(a, r) < 8e-3, (name, rel(a, r))  # tests/test_gpu_configs.py BWD16_TOL


def test_hbm_mirrors_of_destroyed_wrappers_are_reused(ctx):
    """A caller that wraps its arrays per call (umfa.flash_attention_forward does) gets the HBM mirrors of the wrappers it destroyed: no hipMalloc /
    hipFree per array per call; umfa_release_scratch(all) frees what is kept; results do not depend on it."""
    import ctypes

    import umfa
    import umfa_torch
    from umfa import core
    rng = np.random.default_rng(9)
    q, k, v = (bf16_bits(rng, (1, 4, 512, 64)) for _ in range(3))
    kw = dict(input_precision="bf16", intermediate_precision="bf16", layout="bhsd")
    o0 = umfa.flash_attention_forward(ctx, q, k, v, **kw)
    assert core._lib.umfa_release_scratch(ctx.handle, None, 1) == 0  # the cache is empty from here
    h0 = int(umfa_torch.get_option("mirror_cache_hits"))
    o1 = umfa.flash_attention_forward(ctx, q, k, v, **kw)   # four fresh mirrors, handed to the cache on destroy
    assert int(umfa_torch.get_option("mirror_cache_hits")) == h0
    o2 = umfa.flash_attention_forward(ctx, q, k, v, **kw)   # ... and taken again
    assert int(umfa_torch.get_option("mirror_cache_hits")) == h0 + 4
    assert np.array_equal(o1, o0) and np.array_equal(o2, o0)
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        umfa.flash_attention_forward(ctx, q, k, v, **kw)
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (8 << 20)  # flat
    assert core._lib.umfa_release_scratch(ctx.handle, None, 1) == 0
    h1 = int(umfa_torch.get_option("mirror_cache_hits"))
    umfa.flash_attention_forward(ctx, q, k, v, **kw)
    assert int(umfa_torch.get_option("mirror_cache_hits")) == h1
