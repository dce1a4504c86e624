"""GPU parity of the runtime-quantised path (mfa_quantized_forward_with_lse / mfa_quantized_backward,
MFABridge+Quantized.swift:227-533) against the oracle's restatement: quantise Q, K, V with the reference's
symmetric formula (QuantizationTests.swift:72-128), de-quantise, fp64 SDPA."""
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


def rel_err(a, ref):
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("shape", [(1, 2, 128, 64), (2, 2, 200, 128), (1, 1, 70, 40), (1, 2, 256, 256)])
@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("mode", ["tensor", "blockwise"])
@pytest.mark.parametrize("causal", [False, True])
def test_quantized_forward_vs_oracle(ctx, shape, bits, mode, causal):
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(5)
    q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    o, lse = umfa.quantized_attention(ctx, q, k, v, causal=causal, precision=f"int{bits}", quant_mode=mode,
                                      layout="bhsd", return_lse=True)
    assert ctx.last_kernel.startswith("fa_fwd_i")
    ref, rlse = orc.quantized_forward(q, k, v, causal=causal, bits=bits, quant_mode=0 if mode == "tensor" else 2)
    assert np.isfinite(o).all()
    # same quantised operands on both sides; the GPU rounds P and the de-quantised V to fp16
    assert rel_err(o, ref) < 2e-3, rel_err(o, ref)
    assert np.abs(lse.reshape(rlse.shape) - rlse).max() < 2e-3
    # and the quantisation error itself stays inside the reference's budget for the format
    exact = orc.sdpa_forward(q, k, v, causal=causal)
    assert rel_err(o, exact) < (0.08 if bits == 8 else 0.9)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_quantized_forward_16bit_inputs_and_mask(ctx, dt):
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(6)
    shape = (1, 2, 96, 64)
    f = [rng.standard_normal(shape).astype(np.float32) for _ in range(3)]
    if dt == "fp16":
        q, k, v = (a.astype(np.float16) for a in f)
    else:
        q, k, v = (orc.f32_to_bf16_bits(a).reshape(shape) for a in f)
    mask = (rng.standard_normal((1, 2, 96, 96)) * 2).astype(np.float32)
    o = umfa.quantized_attention(ctx, q, k, v, precision="int8", quant_mode="blockwise", layout="bhsd",
                                 attn_mask=mask, input_precision=dt)
    ref, _ = orc.quantized_forward(q, k, v, mask=mask, bits=8, quant_mode=2)
    assert rel_err(np.asarray(o, np.float32), ref) < (4e-3 if dt == "fp16" else 2e-3)


def test_quantized_backward_vs_oracle(ctx):
    """Gradients of the de-quantised operands (STE), fp32 out; reference thresholds for the quantised
    backward are cosine >= 0.7 / rel-err <= 30 % (docs/attic/QUANTIZED_TRAINING_BINDINGS.md:14,75) --
    against the oracle on the SAME de-quantised operands we hold 1e-4."""
    import ctypes
    import umfa
    from umfa._ffi import _lib
    orc = _oracle()
    rng = np.random.default_rng(7)
    B, H, S, D = 1, 2, 128, 64
    q, k, v, do = (rng.standard_normal((B, H, S, D)).astype(np.float32) for _ in range(4))
    o, lse = umfa.quantized_attention(ctx, q, k, v, precision="int8", quant_mode="blockwise", layout="bhsd",
                                      return_lse=True)
    dq, dk, dv = (np.zeros((B, H, S, D), np.float32) for _ in range(3))
    bufs = [umfa.MFABuffer(ctx, a) for a in (q, k, v, o, do, lse, dq, dk, dv)]
    rc = _lib.mfa_quantized_backward(ctx.handle, *(b.handle for b in bufs), None, B, S, S, H, D,
                                     1.0 / np.sqrt(D), False, 3, 2, 2)
    for b in bufs:
        b.close()
    assert rc == 0
    assert ctx.last_kernel.startswith("fa_bwd16<fp16"), ctx.last_kernel  # the 16-bit MFMA backward on fp16 de-quantised operands
    # oracle: de-quantise with the same formula, then the fp64 backward
    def fake(x):
        out = np.empty_like(x)
        for h in range(H):
            qv, sc = orc.quantize_symmetric(x[0, h], group=64 * D)
            out[0, h] = orc.dequantize(qv, sc, group=64 * D).reshape(S, D)
        return out
    fq, fk, fv = fake(q), fake(k), fake(v)
    ro, rlse = orc.sdpa_forward(fq, fk, fv, return_lse=True)
    rdq, rdk, rdv, _ = orc.sdpa_backward(do, fq, fk, fv, o, lse.reshape(B, H, S))
    for got, ref, name in [(dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")]:
        assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), (name, np.abs(got - ref).max())
        cos = float((got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum()))
        assert cos > 0.999


@pytest.mark.parametrize("shape", [(1, 2, 256, 64), (1, 2, 256, 128), (2, 2, 512, 448), (1, 3, 1280, 1100), (1, 2, 768, 768),
                                   (2, 24, 1024, 1024)])  # the last one: > 256 x 1 tile steps, i.e. real multi-tile loops
@pytest.mark.parametrize("bits,mode", [(8, "blockwise"), (8, "tensor"), (4, "blockwise")])
@pytest.mark.parametrize("causal", [False, True])
def test_quantized_forward_w64_vs_oracle(ctx, shape, bits, mode, causal):
    """head_dim 128 shapes that take the 64-rows-per-wave quantised kernel (fa_fwd_w64_i8): one / two / many key tiles,
    items cut into parts, ragged Sq and Skv, causal; same operands and tolerances as the 128-row kernel above"""
    import umfa
    orc = _oracle()
    B, H, Sq, Skv = shape
    rng = np.random.default_rng(Sq + Skv + bits)
    q = rng.standard_normal((B, H, Sq, 128)).astype(np.float32)
    k = rng.standard_normal((B, H, Skv, 128)).astype(np.float32)
    v = rng.standard_normal((B, H, Skv, 128)).astype(np.float32)
    o, lse = umfa.quantized_attention(ctx, q, k, v, causal=causal, precision=f"int{bits}", quant_mode=mode,
                                      layout="bhsd", return_lse=True)
    assert ctx.last_kernel.startswith("fa_fwd_w64_i"), ctx.last_kernel
    ref, rlse = orc.quantized_forward(q, k, v, causal=causal, bits=bits, quant_mode=0 if mode == "tensor" else 2)
    assert np.isfinite(o).all()
    assert rel_err(o, ref) < 2e-3, rel_err(o, ref)
    assert np.abs(lse.reshape(rlse.shape) - rlse).max() < 2e-3
    o2, _ = umfa.quantized_attention(ctx, q, k, v, causal=causal, precision=f"int{bits}", quant_mode=mode, layout="bhsd",
                                     return_lse=True)
    assert np.array_equal(o, o2)  # bitwise reproducible, including the fold of cut items


@pytest.mark.parametrize("gain,outlier", [(1.0, 60.0), (3.0, 0.0), (0.05, 0.0)])
def test_quantized_forward_w64_wide_dynamic_range(ctx, gain, outlier):
    """The int8 kernel keeps its integer scores in the mantissa of a biased float (no int -> float conversion per score);
    the bias is cancelled exactly by rounding the per-tile softmax scale to ~22 bits.  Large block scales (outliers,
    large activations: BIAS * c2 up to ~1e6) and tiny ones must stay inside the same tolerance as N(0,1) data."""
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(int(gain * 100) + int(outlier))
    B, H, Sq, Skv = 1, 2, 512, 1024
    q = (rng.standard_normal((B, H, Sq, 128)) * gain).astype(np.float32)
    k = (rng.standard_normal((B, H, Skv, 128)) * gain).astype(np.float32)
    v = rng.standard_normal((B, H, Skv, 128)).astype(np.float32)
    if outlier:
        q[0, :, ::64, 5] = outlier    # one outlier per quantisation block: block scale x15, ordinary logits
        k[0, :, 17::128, 9] = -outlier  # every second K block only: neighbouring tiles with very different scales
    o, lse = umfa.quantized_attention(ctx, q, k, v, precision="int8", quant_mode="blockwise", layout="bhsd", return_lse=True)
    assert ctx.last_kernel.startswith("fa_fwd_w64_i"), ctx.last_kernel
    ref, rlse = orc.quantized_forward(q, k, v, bits=8, quant_mode=2)
    assert np.isfinite(o).all()
    assert rel_err(o, ref) < 2e-3, rel_err(o, ref)
    assert np.abs(lse.reshape(rlse.shape) - rlse).max() < 2e-3 * max(1.0, np.abs(rlse).max() / 50)


@pytest.mark.parametrize("kind", ["ramp_up", "ramp_down", "jump", "causal_shift"])
def test_quantized_forward_w64_moving_reference(ctx, kind):
    """The int8 kernel runs the lazy softmax reference with fp16 P (rebase at 2^6, give up at 2^15): scores that rise along
    the key axis (many rebases), fall (none), jump by hundreds of nats inside one tile (overflow: the segment re-runs with the
    max chain), and a causal sweep whose scores all sit far below zero (rows that start on the reference 0: the row-sum floor
    of the fp16 mode re-runs them) -- against the oracle on the same quantised operands, and bit-equal to the deferred mode's
    tolerance class"""
    import umfa
    orc = _oracle()
    rng = np.random.default_rng(77)
    B, H, Sq, Skv = 1, 2, 512, 1024
    q = rng.standard_normal((B, H, Sq, 128)).astype(np.float32)
    k = (rng.standard_normal((B, H, Skv, 128)) * 0.3).astype(np.float32)
    v = rng.standard_normal((B, H, Skv, 128)).astype(np.float32)
    d = q.mean(axis=2, keepdims=True)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    ramp = np.linspace(0.0, 1.0, Skv, dtype=np.float32).reshape(1, 1, Skv, 1)
    causal = False
    if kind == "ramp_up":
        k = k + ramp * 60.0 * 11.3 * d
    elif kind == "ramp_down":
        k = k - ramp * 60.0 * 11.3 * d
    elif kind == "jump":
        k[:, 1, 300:] += 4000.0 * d[:, 1]
    else:
        causal = True
        q[:, 1, :, 0] = np.abs(q[:, 1, :, 0]) + 1.5
        k[:, 1, :, 0] -= 300.0
    o, lse = umfa.quantized_attention(ctx, q, k, v, causal=causal, precision="int8", quant_mode="blockwise", layout="bhsd", return_lse=True)
    assert ctx.last_kernel.startswith("fa_fwd_w64_i8<"), ctx.last_kernel
    ref, rlse = orc.quantized_forward(q, k, v, causal=causal, bits=8, quant_mode=2)
    assert np.isfinite(o).all()
    assert rel_err(o, ref) < 2e-3, (kind, rel_err(o, ref))
    fin = np.isfinite(rlse)
    assert np.abs(lse.reshape(rlse.shape) - rlse)[fin].max() < 2e-3 * max(1.0, np.abs(rlse[fin]).max() / 50)
    o2, _ = umfa.quantized_attention(ctx, q, k, v, causal=causal, precision="int8", quant_mode="blockwise", layout="bhsd", return_lse=True)
    assert np.array_equal(o, o2)


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("bits,mode", [(8, 2), (8, 0), (4, 2)])
def test_quantiser_is_bit_exact_with_the_oracle(dt, bits, mode):
    """integer work: the int8 / int4 values and the fp32 scales of the runtime quantiser must equal the oracle's
    (QuantizationTests.swift:72-128: scale = absmax / qmax, q = clamp(round-half-away(x / scale))) bit for bit,
    including values that sit exactly on or next to a rounding boundary"""
    import ctypes
    import umfa_torch
    from umfa._ffi import _lib, _check_error
    from umfa_torch import ops
    orc = _oracle()
    torch.manual_seed(bits + mode)
    BH, S, D = 6, 200, 128
    x = torch.randn(BH, S, D, device="cuda")
    # plant boundary cases: exact multiples of scale/2 of the first block once its scale is absmax / qmax
    qmax = 127.0 if bits == 8 else 7.0
    x[0, 0, 0] = 4.0                                   # becomes the block absmax -> scale = 4 / qmax exactly
    x[0, 1, :16] = torch.arange(16, device="cuda") * (4.0 / qmax) + (2.0 / qmax)   # k + 0.5 steps: ties
    x[0, 2, :16] = -x[0, 1, :16]
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dt]
    xt = x.to(tdt).contiguous()
    q8 = torch.empty(BH * S * 128, dtype=torch.int8, device="cuda")
    sc = torch.empty(BH * ((S + 63) // 64), dtype=torch.float32, device="cuda")
    pad = ctypes.c_uint32(0)
    stream = torch.cuda.current_stream().cuda_stream
    _check_error(_lib.umfa_quantize_rows(ops.context(), ctypes.c_void_p(stream), ctypes.c_void_p(xt.data_ptr()),
                                         {"fp16": 0, "bf16": 1, "fp32": 2}[dt], BH, S, D, bits, mode,
                                         ctypes.c_void_p(q8.data_ptr()), ctypes.c_void_p(sc.data_ptr()), ctypes.byref(pad)))
    torch.cuda.synchronize()
    assert pad.value == 128
    got_q = q8.cpu().numpy().reshape(BH, S, 128)
    got_s = sc.cpu().numpy().reshape(BH, -1)
    xf = xt.float().cpu().numpy()
    if mode == 2:  # one scale per 64-row block of a (batch, head) slab
        for bh in range(BH):
            for b0 in range(0, S, 64):
                ref_q, ref_s = orc.quantize_symmetric(xf[bh, b0:b0 + 64], bits=bits)
                assert np.array_equal(got_q[bh, b0:b0 + 64].ravel(), ref_q), (bh, b0)
                assert got_s[bh, b0 // 64] == ref_s[0]
    else:          # one scale for the whole tensor
        ref_q, ref_s = orc.quantize_symmetric(xf, bits=bits)
        assert np.array_equal(got_q.ravel(), ref_q)
        assert (got_s == ref_s[0]).all()


@pytest.mark.parametrize("shape,causal,bits,mode", [((1, 24, 1024, 128), False, 8, "blockwise"), ((2, 3, 333, 64), True, 8, "tensor"),
                                                    ((1, 2, 512, 128), False, 4, "blockwise"), ((1, 2, 200, 256), False, 8, "blockwise")])
def test_quantized_forward_stream_entry_matches_blocking(shape, causal, bits, mode):
    """umfa_quantized_forward_stream (asynchronous, caller's stream) gives the bits of mfa_quantized_forward_with_lse"""
    import umfa_torch
    torch.manual_seed(8)
    q, k, v = (torch.randn(shape, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o1, l1 = umfa_torch.quantized_attention_forward(q, k, v, causal=causal, bits=bits, quant_mode=mode)
    k1 = umfa_torch.last_kernel()
    o2, l2 = umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, bits=bits, quant_mode=mode, return_lse=True)
    torch.cuda.synchronize()
    assert umfa_torch.last_kernel() == k1
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    m = torch.rand(shape[0], shape[1], shape[2], shape[2], device="cuda") < 0.8
    m[..., 0] = True
    if not causal:
        a, _ = umfa_torch.quantized_attention_forward(q, k, v, mask=m, bits=bits, quant_mode=mode)
        ka = umfa_torch.last_kernel()
        b = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, bits=bits, quant_mode=mode)
        torch.cuda.synchronize()
        # the blocking entry reads the reference ABI's dense fp32 expansion (fa_fwd_i8); the in-stream one takes the caller's bool tensor and, from round 6
        # on, shapes with enough 256-row blocks run the one-wave-per-SIMD int8 kernel's mask instantiation: another kernel, the same answer to its tolerance
        if umfa_torch.last_kernel() == ka:
            assert torch.equal(a, b)
        else:
            assert umfa_torch.last_kernel().endswith(",mask>") and float((a - b).abs().max()) <= 2e-3 * float(a.abs().max())


@pytest.mark.parametrize("D,dt", [(128, "bf16"), (64, "fp16"), (80, "bf16")])
def test_quantized_backward_stream_entry_matches_the_blocking_entry(ctx, D, dt):
    """umfa_quantized_backward_stream (in-stream, raw device pointers) = mfa_quantized_backward's numbers: the 16-bit MFMA
    engine at head_dim 64 / 128 (status stays 0), the fp32-exact engine at head_dim 80; and a V beyond fp16's range goes through the fp16
    engine like any other (every operand enters it as a power-of-two multiple)."""
    import ctypes
    import torch
    import umfa_torch
    from umfa._ffi import _lib
    from umfa_torch import ops
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    torch.manual_seed(13)
    B, H, S = 1, 2, 256
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=tdt) for _ in range(4))
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, return_lse=True)
    dq, dk, dv, status = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse)
    torch.cuda.synchronize()
    kern = umfa_torch.last_kernel()
    assert kern.startswith("fa_bwd16<fp16") if D != 80 else kern.startswith("fa_bwd_exact"), kern
    assert int(status.item()) == 0
    # the blocking entry on the same device tensors (zero-copy wraps)
    gq, gk, gv = (torch.zeros_like(t) for t in (dq, dk, dv))

    def wrap(t):
        from umfa import _ffi
        h = _ffi.mfa_buffer_t()
        _ffi._check_error(_lib.mfa_buffer_from_mtl_buffer(ops.context(), ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(), ctypes.byref(h)))
        return h
    hs = [wrap(t) for t in (q, k, v, o, do, lse, gq, gk, gv)]
    rc = _lib.mfa_quantized_backward(ops.context(), *hs, None, B, S, S, H, D, float(D) ** -0.5, False, 3, 2, ops._PREC[tdt])
    assert rc == 0
    for h in hs:
        _lib.mfa_destroy_buffer(h)
    for a, b in ((dq, gq), (dk, gk), (dv, gv)):
        assert torch.equal(a, b)
    if D != 80:
        big = (v.float() * 3.0e5).to(tdt) if dt == "bf16" else None
        if big is not None:  # bf16 V beyond 65504: the fp16 engine takes it as a power-of-two multiple (round 5; a status of 1 and no gradients before)
            o2, lse2 = umfa_torch.quantized_attention_forward_stream(q, k, big, return_lse=True)
            g2 = umfa_torch.quantized_attention_backward_stream(do, q, k, big, o2, lse2)
            with umfa_torch.options(bwd_exact=1):
                r2 = umfa_torch.quantized_attention_backward_stream(do, q, k, big, o2, lse2)
            torch.cuda.synchronize()
            assert int(g2[3].item()) == 0
            for a, b in zip(g2[:3], r2[:3]):
                assert torch.isfinite(a).all()
                assert float((a.double() - b.double()).abs().max() / b.double().abs().max()) < 2e-3


@pytest.mark.parametrize("D,dt", [(128, "bf16"), (64, "bf16"), (128, "fp16")])
@pytest.mark.parametrize("gain", [1e-9, 1e-6, 1e-3, 1e3])
def test_quantized_backward_over_the_range_of_dout(D, dt, gain):
    """mfa_quantized_backward's engine (fp16 images of the de-quantised operands, 16-bit MFMA backward) with the dO of a real training
    step: 1e-6, 1e-9 -- as a plain fp16 cast those were subnormals / zeros and the gradients were off by 35 % / all zero with status 0
    (tools/lab/qbwd_range_probe.py).  dO goes in as dO * 2^-e, e chosen on the device; against the fp32-exact engine on the same
    inputs: the bound the engine meets at |dO| ~ 1 (fp16 P and dS), status 0."""
    import torch
    import umfa_torch
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    if dt == "fp16" and gain < 1e-6:
        pytest.skip("below fp16's own range: the CALLER's dO is already zero")
    torch.manual_seed(17)
    B, H, S = 1, 3, 512
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=tdt) for _ in range(4))
    do = (do.float() * gain).to(tdt)
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, return_lse=True)
    with umfa_torch.options(bwd_exact=1):
        ref = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse)
        assert umfa_torch.last_kernel().startswith("fa_bwd_exact")
    got = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse)
    torch.cuda.synchronize()
    assert umfa_torch.last_kernel().startswith("fa_bwd16<fp16"), umfa_torch.last_kernel()
    assert int(got[3].item()) == 0
    for a, b, name in zip(got[:3], ref[:3], ("dq", "dk", "dv")):
        a, b = a.double(), b.double()
        assert torch.isfinite(a).all(), name
        assert float((a - b).abs().max() / b.abs().max()) < 2e-3, (name, float((a - b).abs().max() / b.abs().max()))


@pytest.mark.parametrize("gain", [1e-9, 1e-5, 3e4, 1e7, 1e20])
@pytest.mark.parametrize("shape,mode,bits,dt", [((1, 3, 1024, 1024, 128), "blockwise", 8, "bf16"), ((2, 2, 512, 448, 128), "blockwise", 4, "fp16"),
                                               ((1, 2, 768, 768, 128), "tensor", 8, "bf16"), ((1, 2, 16640, 16640, 128), "blockwise", 8, "bf16"),
                                               ((1, 3, 200, 333, 64), "blockwise", 8, "bf16"), ((2, 2, 128, 256, 80), "tensor", 8, "fp32"),
                                               ((1, 2, 1024, 512, 128), "blockwise", 8, "fp32"), ((1, 2, 512, 512, 64), "blockwise_fp8pv", 8, "bf16"),
                                               ((1, 2, 1024, 1024, 128), "blockwise_fp8pv", 8, "bf16")])
def test_quantized_forward_over_the_range_of_v(gain, shape, mode, bits, dt):
    """The forward's P V product runs in fp16 on an fp16 image of the de-quantised V; fp16 has five exponent bits.  As plain q * s the
    image was inf for |v| ~ 1e5 and subnormal below ~ 1e-4 (tools/lab/qfwd_range_probe.py: NaN / 1.5 % / all zero, no error).  It is
    q * s * 2^-e now, one power of two per (batch, head) slab found on the device -- by an exchange among the wave quantiser's workgroups
    (16-bit operands, head_dim 64 / 128, block-wise), or by a pass of its own (tensor-wise mode, fp32 operands, other head dims, slabs of
    more than 256 blocks: the 16640-key case) -- and 2^e comes back in the kernels' epilogues.  Oracle parity per slab at every scale, with
    one slab 1e6 times smaller than its neighbours on top."""
    import torch
    import umfa_torch
    orc = _oracle()
    B, H, Sq, Skv, D = shape
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dt]
    if dt == "fp16" and not 1e-5 <= gain <= 1e4:
        pytest.skip("outside fp16: the caller's V cannot hold it")
    torch.manual_seed(Sq + Skv + D + bits)
    q = torch.randn(B, H, Sq, D, device="cuda").to(tdt)
    k = torch.randn(B, H, Skv, D, device="cuda").to(tdt)
    v = torch.randn(B, H, Skv, D, device="cuda") * gain
    if mode != "tensor" and dt != "fp16":
        v[0, H - 1] *= 1e-6  # (a tensor-wide scale would quantise this slab to zero: the quantiser's business, not the image's)
    v = v.to(tdt)
    o = umfa_torch.quantized_attention_forward_stream(q, k, v, bits=bits, quant_mode=mode)
    kern = umfa_torch.last_kernel()
    o2 = umfa_torch.quantized_attention_forward_stream(q, k, v, bits=bits, quant_mode=mode)
    torch.cuda.synchronize()
    assert torch.equal(o, o2)
    f32 = lambda t: t.float().cpu().numpy()
    ref, _ = orc.quantized_forward(f32(q), f32(k), f32(v), bits=bits, quant_mode=0 if mode == "tensor" else 2)
    o = o.cpu().numpy().astype(np.float64)
    assert np.isfinite(o).all(), kern
    tol = 6e-2 if kern.startswith("fa_fwd_w64_i8f8") else 2e-3
    for b in range(B):
        for h in range(H):
            top = np.abs(ref[b, h]).max()
            assert top > 0
            assert np.abs(o[b, h] - ref[b, h]).max() < tol * top, (b, h, np.abs(o[b, h] - ref[b, h]).max() / top, kern)


def test_quantized_forward_v_exchange_without_waiting():
    """cast_wait_us = 0: a workgroup of the wave quantiser that is not served at once reads the slab's amax itself; cast_two_pass: the
    amax pass of its own; quant_block_wg: the workgroup-per-block quantiser (same pass).  Same exponent every way: bit-identical O."""
    import torch
    import umfa_torch
    torch.manual_seed(5)
    q, k = (torch.randn(1, 4, 1024, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    v = (torch.randn(1, 4, 1024, 128, device="cuda") * 1e-7).to(torch.bfloat16)
    outs = []
    for opts in ({}, {"cast_wait_us": 0}, {"cast_two_pass": 1}, {"quant_block_wg": 1}, {}):
        with umfa_torch.options(**opts):
            outs.append(umfa_torch.quantized_attention_forward_stream(q, k, v))
            assert umfa_torch.last_kernel() == "fa_fwd_w64_i8<128>"
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("gq,gk,gv", [(1e-4, 1e4, 1e-7), (3e2, 3e-3, 1e6), (1.0, 1.0, 1e12), (1e-3, 1e3, 1.0)])
@pytest.mark.parametrize("D,causal", [(128, False), (64, True), (256, False)])
def test_quantized_backward_over_the_range_of_every_operand(D, causal, gq, gk, gv):
    """Q, K, V of any magnitude (their product Q K^T kept where a softmax makes sense): the quantiser's fp16 copies are q * s * 2^-e with the
    tensor's largest magnitude in [1, 2), dO likewise, the exponents come back through BwdParams::units (softmax scale, D, the three
    gradients).  Against the fp32-exact engine on the same inputs; status 0; blocking entry = in-stream entry."""
    import torch
    import umfa_torch
    torch.manual_seed(23)
    B, H, S = 1, 2, 384
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda") for _ in range(4))
    q, k, v, do = (q * gq).bfloat16(), (k * gk).bfloat16(), (v * gv).bfloat16(), (do * 1e-6).bfloat16()
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, return_lse=True)
    with umfa_torch.options(bwd_exact=1):
        ref = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse, causal=causal)
    got = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse, causal=causal)
    torch.cuda.synchronize()
    assert umfa_torch.last_kernel().startswith("fa_bwd16<fp16"), umfa_torch.last_kernel()
    assert int(got[3].item()) == 0
    for a, b, name in zip(got[:3], ref[:3], ("dq", "dk", "dv")):
        a, b = a.double(), b.double()
        assert torch.isfinite(a).all(), name
        assert float((a - b).abs().max() / b.abs().max()) < 2e-3, (name, float((a - b).abs().max() / b.abs().max()))


@pytest.mark.parametrize("B,H,S,D,causal", [(1, 4, 1024, 128, False), (2, 3, 512, 64, True), (1, 2, 768, 128, True), (1, 1, 320, 256, False)])
def test_quantized_forward_and_backward_replayed_in_a_graph_follow_the_data(B, H, S, D, causal):
    """ONE captured quantised forward + backward, replayed while V and dO change scale by 2^±20 between replays -- largest magnitudes going DOWN as well as
    up: the exponents of the fp16 images are found on the device inside the captured launches (amax passes, the quantiser's exchange), nothing is baked
    in at capture and nothing is left behind by the previous replay -- every replay equals the eager call on the same data, bit for bit.  (The amax words
    of the backward are updated with agent-scope atomics; since round 6 they sit in a self-cleaning block of their own and no memset node precedes them:
    runtime_internal.h StreamScratch::ensure_qhdr.  A stale word would act as max(previous, current) and show up here on the steps that shrink.)"""
    import torch
    import umfa_torch
    torch.manual_seed(31)
    q, k = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    v0, do0 = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    v, do = v0.clone(), do0.clone()
    out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    lse = torch.empty(B * H * S, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):  # warm the (device, stream) pools on the stream that will capture
            umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, return_lse=True, out=out, lse=lse)
            g_eager = umfa_torch.quantized_attention_backward_stream(do, q, k, v, out, lse, causal=causal)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, return_lse=True, out=out, lse=lse)
            g_cap = umfa_torch.quantized_attention_backward_stream(do, q, k, v, out, lse, causal=causal)
        for ev, edo in ((0, 0), (20, -20), (-20, 20), (0, -30), (-25, -35), (3, 0)):
            v.copy_((v0.float() * 2.0 ** ev).to(torch.bfloat16))
            do.copy_((do0.float() * 2.0 ** edo).to(torch.bfloat16))
            graph.replay()
            side.synchronize()
            o_r, grads_r = out.clone(), [t.clone() for t in g_cap[:3]]
            o_e = umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, return_lse=True)
            g_e = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o_e[0], o_e[1], causal=causal)
            side.synchronize()
            assert torch.equal(o_r, o_e[0]), (ev, edo)
            for a, b in zip(grads_r, g_e[:3]):
                assert torch.isfinite(a).all() and torch.equal(a, b), (ev, edo)
            assert int(g_cap[3].item()) == 0
            # and linear in the scales, to the engine's tolerance
            if (ev, edo) == (0, 0):
                base = [t.double() for t in grads_r]
            else:
                for a, b0, e in zip(grads_r, base, (ev + edo, ev + edo, edo)):
                    assert float((a.double() * 2.0 ** -e - b0).abs().max() / b0.abs().max()) < 3e-3, (ev, edo)


@pytest.mark.parametrize("kind", ["bool_2d", "bool_padding", "bool_per_head", "f16_3d", "f32_strided", "bf16_bias", "f32_dense"])
@pytest.mark.parametrize("shape,causal", [((2, 3, 300, 333, 128), False), ((1, 2, 512, 512, 64), False), ((1, 2, 257, 257, 128), True)])
def test_quantized_forward_with_the_callers_mask_tensor(kind, shape, causal):
    """umfa_quantized_forward_masked_stream: the mask as the caller has it -- any <= 4-D broadcastable bool / fp16 / bf16 / fp32 tensor, read in
    place with its strides (mfa_prepare_mask's semantics, MFABridge.swift:157-242) -- instead of the dense fp32 [B, H, Sq, Skv] expansion the
    reference's quantised entry takes.  Against the oracle's quantised restatement with the expanded mask, and against this library's own dense
    entry on the expansion (the same kernel reading different bytes)."""
    import ctypes
    import torch
    import umfa_torch
    from umfa_torch import ops
    orc = _oracle()
    B, H, Sq, Skv, D = shape
    torch.manual_seed(Sq + D)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    g = torch.Generator(device="cuda").manual_seed(7)
    if kind == "bool_2d":
        m = torch.rand(Sq, Skv, device="cuda", generator=g) < 0.5
        m[:, 0] = True
        m[5] = False  # a row that attends to nothing
    elif kind == "bool_padding":
        m = (torch.arange(Skv, device="cuda")[None, None, None, :] < torch.tensor([Skv - 40 * (b + 1) for b in range(B)], device="cuda")[:, None, None, None])
    elif kind == "bool_per_head":
        m = torch.rand(1, H, Sq, Skv, device="cuda", generator=g) < 0.7
        m[..., 3] = True
    elif kind == "f16_3d":
        m = (torch.randn(H, Sq, Skv, device="cuda", generator=g) * 2).half()
    elif kind == "f32_strided":
        m = (torch.randn(B, 1, Sq, 2 * Skv, device="cuda", generator=g) * 2)[..., ::2]
    elif kind == "bf16_bias":
        m = (torch.randn(1, 1, Sq, Skv, device="cuda", generator=g) * 3).bfloat16()
        m[0, 0, :, Skv // 2:] = float("-inf")
    else:
        m = torch.randn(B, H, Sq, Skv, device="cuda", generator=g)
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, causal=causal, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern.startswith("fa_fwd_i8<"), kern
    full = torch.zeros(B, H, Sq, Skv, device="cuda", dtype=torch.float32)
    full = full.masked_fill(~m, float("-inf")) if m.dtype == torch.bool else full + m.float()
    full = full.contiguous()
    ref, rlse = orc.quantized_forward(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), causal=causal,
                                      mask=full.cpu().numpy(), bits=8, quant_mode=2)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    assert rel_err(on, ref) < 2e-3, rel_err(on, ref)
    dead = np.isneginf(rlse)
    assert (on.reshape(-1, D)[dead.reshape(-1)] == 0).all()
    # the reference ABI's form of the same mask through the dense entry
    o2 = torch.empty_like(o)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    rc = ops._lib.umfa_quantized_forward_stream(ops.context(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), vp(q), vp(k), vp(v), vp(o2), None,
                                                vp(full), B, Sq, Skv, H, D, float(D) ** -0.5, causal, 3, 2, ops._PREC[q.dtype])
    assert rc == 0
    torch.cuda.synchronize()
    if m.dtype == torch.bool:
        assert torch.equal(o, o2)  # 0 / -inf terms: the same arithmetic whichever bytes they were read from
    else:
        # (16-byte loads fuse the log2(e) multiply into an fma, scalar reads do not: a last-bit difference in a score can flip the fp16 rounding of its P)
        assert float((o - o2).abs().max()) <= 2e-4 * float(o2.abs().max())


def test_quantized_forward_tensorwise_slab_just_above_half_a_step():
    """tensor-wise mode, fp32 operands: a slab whose largest |v| is just above s / 2 quantises to q = +-1, so its image entries q s are almost TWICE the
    slab's amax -- with the amax at the top of a binade the image needs its second binade of headroom (vimage_exponent) or it rounds to inf."""
    import torch
    import umfa_torch
    orc = _oracle()
    torch.manual_seed(3)
    B, H, S, D = 1, 2, 256, 128
    q, k = (torch.randn(B, H, S, D, device="cuda") for _ in range(2))
    v = torch.zeros(B, H, S, D, device="cuda")
    step = 3.9992
    v[0, 0] = torch.randn(S, D, device="cuda")
    v[0, 0, 7, 3] = 127.0 * step                      # the tensor's amax: s = 3.9992
    v[0, 1] = torch.where(torch.rand(S, D, device="cuda") < 0.5, 1.9998, -1.9998)  # slab 1: |v| = 1.9998 > s / 2 everywhere -> q = +-1, q s = 3.9992
    o = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="tensor")
    ref, _ = orc.quantized_forward(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), bits=8, quant_mode=0)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    for h in range(H):
        assert np.abs(on[0, h] - ref[0, h]).max() < 2e-3 * np.abs(ref[0, h]).max(), h


def test_quantized_entries_on_three_streams_at_once():
    """the quantised forward (V exchange among the quantiser's workgroups, slab headers of the stream's pool) and backward (amax passes, units table in the
    stream's workspace) running on three streams at the same time, different shapes and scales: every stream's results equal its own serial run, bit for bit"""
    import torch
    import umfa_torch
    torch.manual_seed(77)
    cases = []
    for i, (H, S, D, ev) in enumerate([(8, 1024, 128, -18), (4, 2048, 128, 0), (6, 1024, 64, 25)]):
        q, k, v, do = (torch.randn(1, H, S, D, device="cuda") for _ in range(4))
        cases.append([t.bfloat16() for t in (q, k, v * 2.0 ** ev, do * 2.0 ** (-ev - 10))])
    serial = []
    for q, k, v, do in cases:
        o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, return_lse=True)
        g = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse)
        serial.append((o.clone(), [t.clone() for t in g[:3]]))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in cases]
    for rep in range(6):
        outs = []
        for s, (q, k, v, do) in zip(streams, cases):
            with torch.cuda.stream(s):
                o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, return_lse=True)
                g = umfa_torch.quantized_attention_backward_stream(do, q, k, v, o, lse)
                outs.append((o, g))
        torch.cuda.synchronize()
        for (o, g), (so, sg) in zip(outs, serial):
            assert torch.equal(o, so)
            assert int(g[3].item()) == 0
            for a, b in zip(g[:3], sg):
                assert torch.equal(a, b)


def test_quantized_forward_with_the_callers_mask_replays_in_a_graph():
    """umfa_quantized_forward_masked_stream captured once and replayed while the MASK's content changes: the tile flags are taken inside the captured launches"""
    import torch
    import umfa_torch
    torch.manual_seed(2)
    B, H, S, D = 1, 4, 1024, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    i = torch.arange(S, device="cuda")
    masks = [((i[:, None] // 256) == (i[None, :] // 256))[None, None].contiguous(), (torch.rand(1, 1, S, S, device="cuda") < 0.5), torch.ones(1, 1, S, S, dtype=torch.bool, device="cuda")]
    masks[1][..., 0] = True
    m = masks[0].clone()
    out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    lse = torch.empty(B * H * S, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, out=out, lse=lse)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, out=out, lse=lse)
        for mk in masks:
            m.copy_(mk)
            g.replay()
            side.synchronize()
            ref = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=mk)
            side.synchronize()
            assert torch.equal(out, ref)


@pytest.mark.parametrize("kind", ["bool_2d", "bool_padding", "bool_per_head", "bool_blockdiag", "bool_all_true", "bool_dead_blocks"])
@pytest.mark.parametrize("shape,grid", [((2, 2, 512, 512), 0), ((1, 3, 1280, 777), 0), ((1, 3, 1280, 1400), 4), ((2, 2, 512, 512), 3)])
@pytest.mark.parametrize("bits", [8, 4])
def test_quantized_forward_mask_tensor_on_the_one_wave_per_simd_kernel(kind, shape, grid, bits, umfa_opts):
    """(round 6) a bool mask tensor on fa_fwd_w64_i8's MASKT instantiation -- packed bit words, per-wave tile classes, the visited-tile list of every 256-row block,
    exactly the 16-bit kernels' machinery (fa_aux.hip mask_pack_kernel / mask_list_kernel): against the oracle's quantised restatement WITH the mask
    (mfa_quantized_forward_with_lse's semantics, MFABridge+Quantized.swift:227-358, mask as mfa_prepare_mask reads it: MFABridge.swift:157-242), against
    fa_fwd_i8 on the same call (option no_w64_mask), rows that see nothing (O = 0, LSE = -inf), cut blocks (forced small grids), bitwise repeatable."""
    import torch
    import umfa_torch
    orc = _oracle()
    umfa_opts(force_w64=1)
    if grid:
        umfa_opts(w64_grid=grid)
    B, H, Sq, Skv = shape
    D = 128
    torch.manual_seed(Sq + Skv + bits)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    g = torch.Generator(device="cuda").manual_seed(11)
    i = torch.arange(Sq, device="cuda")[:, None]
    j = torch.arange(Skv, device="cuda")[None, :]
    if kind == "bool_2d":
        m = torch.rand(Sq, Skv, device="cuda", generator=g) < 0.5
        m[:, 0] = True
        m[5] = False
    elif kind == "bool_padding":
        m = (j[None, None] < torch.tensor([Skv - 40 * (b + 1) for b in range(B)], device="cuda")[:, None, None, None])
    elif kind == "bool_per_head":
        m = torch.rand(1, H, Sq, Skv, device="cuda", generator=g) < 0.7
        m[..., 3] = True
    elif kind == "bool_blockdiag":
        m = ((i // 192) == (j // 160))[None, None]
    elif kind == "bool_all_true":
        m = torch.ones(1, 1, Sq, Skv, dtype=torch.bool, device="cuda")
    else:
        m = torch.rand(B, H, Sq, Skv, device="cuda", generator=g) < 0.6
        m[:, :, 5::17] = False
        m[:, 0, 256:512] = False
        if H > 1:
            m[:, 1] = False
    o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, bits=bits, return_lse=True)
    kern = umfa_torch.last_kernel()
    assert kern == ("fa_fwd_w64_i8<128,mask>" if bits == 8 else "fa_fwd_w64_i4<128,mask>"), kern
    full = torch.zeros(B, H, Sq, Skv, device="cuda", dtype=torch.float32).masked_fill(~m, float("-inf")).contiguous()
    ref, rlse = orc.quantized_forward(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), mask=full.cpu().numpy(), bits=bits, quant_mode=2)
    on = o.cpu().numpy()
    assert np.isfinite(on).all()
    assert rel_err(on, ref) < (2e-3 if bits == 8 else 3e-3), rel_err(on, ref)
    dead = np.isneginf(rlse).reshape(-1)
    assert (on.reshape(-1, D)[dead] == 0).all() and np.isneginf(lse.cpu().numpy().reshape(-1)[dead]).all()
    assert torch.equal(o, umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, bits=bits))
    with umfa_torch.options(no_w64_mask=1, force_w64=0, w64_grid=0):
        o2 = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, bits=bits)
        assert umfa_torch.last_kernel().startswith("fa_fwd_i"), umfa_torch.last_kernel()
    assert float((o - o2).abs().max()) <= 2e-3 * float(o2.abs().max())
