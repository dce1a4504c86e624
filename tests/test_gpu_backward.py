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
    dq, dk, dv, _ = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision=dt)
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


@pytest.mark.parametrize("shape", [(1, 2, 128, 128), (2, 2, 130, 128), (1, 3, 333, 128), (1, 1, 1024, 128)])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_mfma16_vs_oracle(ctx, shape, dt, causal):
    """head_dim 128 with 16-bit operands takes the bf16/fp16 MFMA backward (P and dS rounded to the input type
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
