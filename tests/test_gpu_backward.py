"""GPU parity of mfa_attention_backward (MFABridge.swift:3171-3282) through the C ABI: gradients vs torch
autograd in fp64 (golden) and vs the CPU oracle; fp32 grads out; D scratch = rowsum(dO o O)."""
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


def _oracle():
    from oracle import oracle
    return oracle


@pytest.mark.parametrize("tag", ["dense", "causal"])
def test_backward_golden_fp32(ctx, golden_dir, tag):
    import umfa
    g = np.load(golden_dir / "backward_fp32.npz")
    q, k, v, do = (np.ascontiguousarray(g[f"{n}_{tag}"]) for n in ("q", "k", "v", "do"))
    causal = tag == "causal"
    o, lse = umfa.flash_attention_forward(ctx, q, k, v, causal=causal, input_precision="fp32",
                                          intermediate_precision="fp32", layout="bhsd", return_lse=True)
    assert np.abs(o - g[f"o_{tag}"]).max() < 1e-5
    assert np.abs(lse.reshape(g[f"lse_{tag}"].shape) - g[f"lse_{tag}"]).max() < 1e-4
    dq, dk, dv, dvec = umfa.attention_backward(ctx, do, q, k, v, o, lse, causal=causal, input_precision="fp32")
    assert ctx.last_kernel.startswith("fa_bwd")
    for got, name in [(dq, "dq"), (dk, "dk"), (dv, "dv")]:
        ref = g[f"{name}_{tag}"]
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), name
    assert np.abs(dvec.reshape(o.shape[:3]) - (do * o).sum(-1)).max() < 1e-4


@pytest.mark.parametrize("shape,dt", [((1, 2, 100, 64), "fp32"), ((2, 2, 130, 64), "bf16"), ((1, 3, 70, 32), "fp16"),
                                      ((1, 1, 257, 88), "fp32"), ((1, 2, 96, 128), "fp32")])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_vs_oracle(ctx, shape, dt, causal):
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(3)
    mk = lambda: rng.standard_normal(shape).astype(np.float32)  # noqa: E731
    q, k, v, do = mk(), mk(), mk(), mk()
    if dt == "fp16":
        q, k, v, do = (a.astype(np.float16) for a in (q, k, v, do))
    elif dt == "bf16":
        q, k, v, do = (orc.f32_to_bf16_bits(a).reshape(shape) for a in (q, k, v, do))
    o, lse = orc.sdpa_forward(q, k, v, causal=causal, return_lse=True)
    rdq, rdk, rdv, _ = orc.sdpa_backward(do, q, k, v, o, lse, causal=causal)
    dq, dk, dv, _ = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision=dt,
                                            intermediate_precision="fp32")
    assert ctx.last_kernel.startswith("fa_bwd_exact")
    for got, ref, name in [(dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")]:
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max())


def test_backward_bitwise_reproducible(ctx):
    import umfa
    rng = np.random.default_rng(4)
    q, k, v, do = (rng.standard_normal((1, 2, 200, 64)).astype(np.float32) for _ in range(4))
    o, lse = umfa.flash_attention_forward(ctx, q, k, v, input_precision="fp32", intermediate_precision="fp32",
                                          layout="bhsd", return_lse=True)
    a = umfa.attention_backward(ctx, do, q, k, v, o, lse, input_precision="fp32")
    b = umfa.attention_backward(ctx, do, q, k, v, o, lse, input_precision="fp32")
    for x, y in zip(a, b):
        assert np.array_equal(x, y)  # no atomics: single-owner accumulation


@pytest.mark.parametrize("shape", [(1, 2, 128, 128), (2, 2, 130, 128), (1, 3, 333, 128), (1, 1, 1024, 128),
                                   (1, 2, 128, 64), (2, 2, 130, 64), (1, 3, 333, 64), (1, 1, 1024, 64), (1, 2, 31, 64),
                                   (1, 2, 128, 256), (2, 2, 130, 256), (1, 2, 333, 256), (1, 1, 640, 256)])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_mfma16_vs_oracle(ctx, shape, dt, causal):
    """head_dim 256 / 128 / 64 with 16-bit operands takes the bf16/fp16 MFMA backward (P and dS rounded to the input type
    before their second product, like P in the forward): relative bound instead of the fp32 one.  The reference's
    own gradient thresholds are far looser (cosine >= 0.7, rel-err <= 30 %, docs/attic/QUANTIZED_TRAINING_BINDINGS.md)."""
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(9)
    f = [rng.standard_normal(shape).astype(np.float32) for _ in range(4)]
    if dt == "fp16":
        q, k, v, do = (a.astype(np.float16) for a in f)
    else:
        q, k, v, do = (orc.f32_to_bf16_bits(a).reshape(shape) for a in f)
    o, lse = orc.sdpa_forward(q, k, v, causal=causal, return_lse=True)
    rdq, rdk, rdv, rd = orc.sdpa_backward(do, q, k, v, o, lse, causal=causal)
    dq, dk, dv, dvec = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision=dt)
    assert ctx.last_kernel.startswith("fa_bwd16"), ctx.last_kernel
    tol = 4e-3 if dt == "fp16" else 2e-2
    for got, ref, name in [(dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")]:
        assert np.isfinite(got).all()
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        cos = float((got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum()))
        assert rel < tol and cos > 0.9999, (name, rel, cos)
    assert np.abs(dvec.reshape(rd.shape) - rd).max() < 1e-3
    # intermediate_precision = fp32 keeps the exact kernel available for the same operands
    dq2, dk2, dv2, _ = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision=dt,
                                               intermediate_precision="fp32")
    assert ctx.last_kernel.startswith("fa_bwd_exact")
    for got, ref in [(dq2, rdq), (dk2, rdk), (dv2, rdv)]:
        assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())


def _prequant_case(bits, grouped, blockwise, causal, D=64):
    """caller-side quantisation (QuantizationTests.swift:72-128 formula), expected gradients from the oracle's fp64
    backward on the de-quantised operands (K / V broadcast over their group, gradients summed over it)"""
    from oracle import oracle as orc
    rng = np.random.default_rng(bits * 100 + grouped * 10 + blockwise)
    B, H, Hkv, Sq, Skv = 2, 4, (2 if grouped else 4), 80, 96
    qmax = 127 if bits == 8 else 7
    BS = 32

    def quant(x):
        if blockwise:  # one scale per 32 consecutive rows of a (batch, head) slab
            Bq, Hh, S, Dd = x.shape
            nb = (S + BS - 1) // BS
            scales = np.zeros((Bq, Hh, nb), np.float32)
            qv = np.zeros(x.shape, np.int8)
            for b in range(Bq):
                for h in range(Hh):
                    for j in range(nb):
                        blk = x[b, h, j * BS:(j + 1) * BS]
                        sc = max(np.abs(blk).max() / qmax, 1e-12)
                        scales[b, h, j] = sc
                        qv[b, h, j * BS:(j + 1) * BS] = np.clip(np.round(blk / sc), -qmax - 1, qmax)
            deq = qv.astype(np.float32) * np.repeat(scales, BS, axis=2)[:, :, :S, None]
            return qv, 1.0, scales.ravel(), deq
        sc = np.float32(np.abs(x).max() / qmax)
        qv = np.clip(np.round(x / sc), -qmax - 1, qmax).astype(np.int8)
        return qv, float(sc), None, qv.astype(np.float32) * sc

    q = rng.standard_normal((B, H, Sq, D), dtype=np.float32)
    k = rng.standard_normal((B, Hkv, Skv, D), dtype=np.float32)
    v = rng.standard_normal((B, Hkv, Skv, D), dtype=np.float32)
    dout = rng.standard_normal((B, H, Sq, D), dtype=np.float32)
    (q8, qs, qbs, qd), (k8, ks, kbs, kd), (v8, vs, vbs, vd) = quant(q), quant(k), quant(v)
    g = H // Hkv
    kx, vx = np.repeat(kd, g, axis=1), np.repeat(vd, g, axis=1)
    o, lse = orc.sdpa_forward(qd, kx, vx, causal=causal, return_lse=True)
    dq, dkx, dvx, dvec = orc.sdpa_backward(dout, qd, kx, vx, o, lse, causal=causal)
    dk = dkx.reshape(B, Hkv, g, Skv, D).sum(2)
    dv = dvx.reshape(B, Hkv, g, Skv, D).sum(2)
    raw = (lambda a: orc.pack_int4(a)) if bits == 4 else (lambda a: a)
    return dict(q=raw(q8), k=raw(k8), v=raw(v8), out=o, dout=dout, lse=lse.ravel(), q_scale=qs, k_scale=ks, v_scale=vs,
                q_block_scales=qbs, k_block_scales=kbs, v_block_scales=vbs,
                q_block_size=BS if blockwise else 0, k_block_size=BS if blockwise else 0, v_block_size=BS if blockwise else 0,
                q_precision="int8" if bits == 8 else "int4", k_precision="int8" if bits == 8 else "int4",
                v_precision="int8" if bits == 8 else "int4", causal=causal, num_heads=H, num_kv_heads=Hkv, head_dim=D,
                seq_len_q=Sq, seq_len_kv=Skv, batch_size=B), (dq, dk, dv, dvec)


@pytest.mark.parametrize("bits,grouped,blockwise,causal", [(8, False, False, False), (8, True, True, True), (4, False, True, False),
                                                           (4, True, False, True)])
def test_prequantized_backward_abi(ctx, bits, grouped, blockwise, causal, umfa_opts):
    """mfa_attention_backward_{query,kv}_quantized_ex (mfa_ffi.h:542-624).  Default engine: the 16-bit MFMA backward on
    operands de-quantised to fp16 (exact here: |q| <= 127 times a scale) -- P and dS are rounded to fp16 before their second
    product, so the gradients carry fp16's 2^-11 (bound 2e-3 of the largest gradient).  With `bwd_exact` the fp32-exact
    engine runs: the only error left is fp32 arithmetic against the oracle's fp64 (2e-4)."""
    import umfa
    from umfa.core import prequantized_backward
    kwargs, (dq, dk, dv, dvec) = _prequant_case(bits, grouped, blockwise, causal)
    gq, gk, gv, gd = prequantized_backward(ctx, **kwargs)
    assert ctx.last_kernel.startswith("fa_bwd16<fp16"), ctx.last_kernel
    for got, ref, name in ((gq, dq, "dq"), (gk, dk, "dk"), (gv, dv, "dv"), (gd, dvec.ravel(), "D")):
        assert np.isfinite(got).all(), name
        assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max())
    umfa_opts(bwd_exact=1)
    gq, gk, gv, gd = prequantized_backward(ctx, **kwargs)
    assert ctx.last_kernel.startswith("fa_bwd_exact")
    for got, ref, name in ((gq, dq, "dq"), (gk, dk, "dk"), (gv, dv, "dv"), (gd, dvec.ravel(), "D")):
        assert np.isfinite(got).all(), name
        assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), name


@pytest.mark.parametrize("gain", [1e-9, 1e-6, 3e-4, 1e4])
def test_prequantized_backward_over_the_range_of_dout(ctx, gain):
    """Gradients are linear in dO, and the dO a training step hands over is routinely 1e-6 and below: fp16 subnormals or zero as a plain
    cast.  The fp16 engine takes dO * 2^-e (one power of two per call, from the tensor's largest magnitude, on the device:
    fa_aux.hip launch_cast_f16_unit) and gives 2^e back with the gradients; D, which the query call hands to the caller and the kv call
    takes back, stays in true units.  Same relative bound as at |dO| ~ 1, no fall-back to the exact engine."""
    from umfa.core import prequantized_backward
    kwargs, (dq, dk, dv, dvec) = _prequant_case(8, True, True, False)
    kwargs = dict(kwargs)
    kwargs["dout"] = kwargs["dout"] * np.float32(gain)
    gq, gk, gv, gd = prequantized_backward(ctx, **kwargs)
    assert ctx.last_kernel.startswith("fa_bwd16<fp16"), ctx.last_kernel
    for got, ref, name in ((gq, dq, "dq"), (gk, dk, "dk"), (gv, dv, "dv"), (gd, dvec.ravel(), "D")):
        ref = ref * np.float64(np.float32(gain))
        assert np.isfinite(got).all(), name
        assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), (name, np.abs(got - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("vgain,qgain", [(1.0e5, 1.0), (1.0e-9, 1.0), (1.0e12, 1.0e-6), (1.0, 1.0e7)])
def test_prequantized_backward_scales_outside_fp16s_range(ctx, vgain, qgain):
    """caller-side scales that put q * s far outside fp16's range (|v| ~ 3e5, 1e-9, 1e12; Q scaled against K): until the end of round 5 the fast
    engine raised a device flag for values beyond 65504 and the call repeated on the fp32 engine (and values below ~1e-4 were silently coarse); now
    every operand enters it as a power-of-two multiple (two dequant launches per tensor: amax, then x * 2^-e) and the gradients keep the
    engine's bound with no second engine involved"""
    from umfa.core import prequantized_backward
    kwargs, (dq, dk, dv, dvec) = _prequant_case(8, False, False, False)
    kwargs = dict(kwargs)
    kwargs["v_scale"] = kwargs["v_scale"] * vgain
    kwargs["out"] = kwargs["out"] * np.float32(vgain)        # O = P V scales with V
    kwargs["dout"] = kwargs["dout"] * np.float32(1.0 / vgain)  # keeps dP, D and the gradients of Q and K where they were
    kwargs["q_scale"] = kwargs["q_scale"] * qgain             # Q K^T unchanged: K takes the inverse
    kwargs["k_scale"] = kwargs["k_scale"] / qgain
    gq, gk, gv, gd = prequantized_backward(ctx, **kwargs)
    assert ctx.last_kernel.startswith("fa_bwd16<fp16"), ctx.last_kernel
    # dQ = scale dS K scales with K (1 / qgain), dK with Q (qgain), dV = P^T dO with dO (1 / vgain)
    for got, ref, name in ((gq, dq / qgain, "dq"), (gk, dk * qgain, "dk"), (gv, dv / vgain, "dv"), (gd, dvec.ravel(), "D")):
        assert np.isfinite(got).all(), name
        assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), (name, np.abs(got - ref).max() / np.abs(ref).max())


def test_prequantized_backward_head_dim_256(ctx):
    from umfa.core import prequantized_backward
    kwargs, (dq, dk, dv, dvec) = _prequant_case(8, True, True, True, D=256)
    gq, gk, gv, gd = prequantized_backward(ctx, **kwargs)
    assert ctx.last_kernel == "fa_bwd16<fp16,256>"
    for got, ref, name in ((gq, dq, "dq"), (gk, dk, "dk"), (gv, dv, "dv"), (gd, dvec.ravel(), "D")):
        assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), name


def test_prequantized_backward_legacy_entries_and_errors(ctx):
    """the non-_ex entries (mfa_ffi.h:480-540) = per-tensor scales, equal head counts; NULL handles -> error 1"""
    import umfa
    from umfa._ffi import _lib
    from umfa.core import MFABuffer
    kwargs, (dq, dk, dv, dvec) = _prequant_case(8, False, False, False)
    B, H, Sq, Skv, D = kwargs["batch_size"], kwargs["num_heads"], kwargs["seq_len_q"], kwargs["seq_len_kv"], kwargs["head_dim"]
    gq, gk, gv = np.zeros_like(dq), np.zeros_like(dk), np.zeros_like(dv)
    gd = np.zeros(B * H * Sq, np.float32)
    arrs = [kwargs["q"], kwargs["k"], kwargs["v"], np.ascontiguousarray(kwargs["out"], np.float32), kwargs["dout"], kwargs["lse"],
            gq, gk, gv, gd]
    bufs = [MFABuffer(ctx, np.ascontiguousarray(a)) for a in arrs]
    bq, bk, bv, bo, bdo, bl, bdq, bdk, bdv, bd = (b.handle for b in bufs)
    tail = (kwargs["q_scale"], 0, kwargs["k_scale"], 0, kwargs["v_scale"], 0, 3, 3, 3, False, False, False, False, False)
    try:
        assert _lib.mfa_attention_backward_query_quantized(ctx.handle, bq, bk, bv, bo, bdo, bl, bdq, bd, B, Sq, Skv, H, D, *tail) == 0
        assert _lib.mfa_attention_backward_kv_quantized(ctx.handle, bq, bk, bv, bdo, bl, bd, bdk, bdv, B, Sq, Skv, H, D, *tail) == 0
        assert _lib.mfa_attention_backward_query_quantized(ctx.handle, None, bk, bv, bo, bdo, bl, bdq, bd, B, Sq, Skv, H, D, *tail) == 1
        bad = tail[:-1] + (True,)  # transpose_o
        assert _lib.mfa_attention_backward_kv_quantized(ctx.handle, bq, bk, bv, bdo, bl, bd, bdk, bdv, B, Sq, Skv, H, D, *bad) == 1
    finally:
        for b in bufs:
            b.close()
    assert np.abs(gq - dq).max() < 2e-3 * np.abs(dq).max() and np.abs(gk - dk).max() < 2e-3 * np.abs(dk).max()
    assert np.abs(gv - dv).max() < 2e-3 * np.abs(dv).max()  # the fp16 MFMA engine (see test_prequantized_backward_abi)


@pytest.mark.parametrize("shape,dt", [((1, 2, 256, 128), "bf16"), ((2, 3, 333, 64), "fp16"), ((1, 2, 200, 256), "bf16"),
                                      ((1, 2, 130, 80), "bf16"), ((1, 2, 96, 64), "fp32")])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_stream_entry_matches_blocking_abi(ctx, shape, dt, causal):
    """umfa_attention_backward_stream (in-stream, raw pointers): fp32 gradients are bit-identical to
    mfa_attention_backward; with grads_in_input_type they are those values rounded once (MFMA backward only --
    head_dim 80 and fp32 operands fall back to fp32 gradients + cast inside ops.attention_backward)"""
    import umfa
    import umfa_torch
    from umfa_torch import ops
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dt]
    torch.manual_seed(21)
    q, k, v, do = (torch.randn(shape, device="cuda", dtype=tdt) for _ in range(4))
    o32, lse = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=torch.float32, return_lse=True)
    npy = lambda t: (t.view(torch.int16) if t.dtype == torch.bfloat16 else t).cpu().numpy()  # noqa: E731
    host = [npy(t).view(np.uint16) if t.dtype == torch.bfloat16 else npy(t) for t in (do, q, k, v)]
    rdq, rdk, rdv, _ = umfa.attention_backward(ctx, host[0], host[1], host[2], host[3], o32.cpu().numpy(), lse.cpu().numpy(),
                                               causal=causal, input_precision=dt)
    scale = shape[-1] ** -0.5
    g32 = ops.attention_backward(do, q, k, v, o32, lse, scale=scale, causal=causal, grads_in_input_type=False)
    gty = ops.attention_backward(do, q, k, v, o32, lse, scale=scale, causal=causal, grads_in_input_type=True)
    torch.cuda.synchronize()
    for a, b, c in zip(g32, gty, (rdq, rdk, rdv)):
        ref = torch.from_numpy(np.ascontiguousarray(c)).to(tdt)   # the blocking ABI's fp32 gradients, rounded once
        assert a.dtype == tdt and b.dtype == tdt
        assert torch.equal(a.cpu(), ref) and torch.equal(b.cpu(), ref)


def test_backward_stream_accepts_o_in_operand_type(ctx):
    """the autograd path keeps O in the operand type (no separate fp32 copy): D = rowsum(dO o O) then comes from the
    rounded O; gradients stay within the bf16 tolerance of the fp32-O gradients"""
    import umfa_torch
    from umfa_torch import ops
    torch.manual_seed(3)
    shape = (1, 3, 384, 128)
    q, k, v, do = (torch.randn(shape, device="cuda", dtype=torch.bfloat16) for _ in range(4))
    o32, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    a = ops.attention_backward(do, q, k, v, o32, lse, scale=128 ** -0.5, keep_fp32=True)
    b = ops.attention_backward(do, q, k, v, o32.to(torch.bfloat16), lse, scale=128 ** -0.5, keep_fp32=True)
    for x, y in zip(a, b):
        assert x.dtype == torch.float32 and y.dtype == torch.float32
        assert ((x - y).abs().max() / x.abs().max()).item() < 1e-2


@pytest.mark.parametrize("which", ["1", "2"])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_both_dq_kernels_head_dim_128(ctx, which, causal, umfa_opts):
    """head_dim 128 has two dQ kernels (two workgroups per CU with 32-key tiles; one per CU with 64-key tiles and the pinned
    four-phase pipeline); the launcher picks by causality.  UMFA_BWD_DQ forces one: both must meet the oracle on both kinds
    of launch, ragged sizes included."""
    import umfa
    orc = _oracle()
    umfa_opts(bwd_dq=which)
    for shape in [(1, 2, 256, 128), (2, 1, 333, 128), (1, 1, 1024, 128)]:
        rng = np.random.default_rng(21)
        q, k, v, do = (orc.f32_to_bf16_bits(rng.standard_normal(shape).astype(np.float32)).reshape(shape) for _ in range(4))
        o, lse = orc.sdpa_forward(q, k, v, causal=causal, return_lse=True)
        rdq, rdk, rdv, _ = orc.sdpa_backward(do, q, k, v, o, lse, causal=causal)
        dq, dk, dv, _ = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision="bf16")
        assert ctx.last_kernel.startswith("fa_bwd16")
        for got, ref, name in [(dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")]:
            rel = np.abs(got - ref).max() / np.abs(ref).max()
            assert np.isfinite(got).all() and rel < 2e-2, (which, causal, shape, name, rel)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_ds_store_form_gives_the_same_gradients(dt):
    """lab option bwd_ds_store (5 products: dkdv also stores dS, dQ = scale dS K as a GEMM; measured slower, kept as a lab
    build): the same gradients to one rounding of the output"""
    import umfa_torch
    torch.manual_seed(12)
    q, k, v, do = (torch.randn(2, 3, 512, 128, device="cuda", dtype=dt) for _ in range(4))
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    ref = [t.clone() for t in umfa_torch.attention_backward(do, q, k, v, o, lse, scale=128 ** -0.5)]
    with umfa_torch.options(bwd_ds_store=1):
        got = umfa_torch.attention_backward(do, q, k, v, o, lse, scale=128 ** -0.5)
    # D comes from the stand-alone row-sum kernel here (another summation order than the one fused into bwd16_dq2), and dQ sums
    # the dS values bwd16_dkdv rounds (dP - D out of the MFMA chain) where bwd16_dq2 forms its own: one rounding of the output apart
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    for a, b in zip(got, ref):
        assert float((a.float() - b.float()).abs().max()) <= ulp * float(b.float().abs().max())
    assert torch.equal(got[2], ref[2])  # dV = P^T dO does not see D
