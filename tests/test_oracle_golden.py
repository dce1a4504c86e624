"""Pins the CPU oracle (oracle/sdpa_ref.c) against the golden vectors generated from
torch-CPU SDPA -- the reference's own declared ground truth."""
import numpy as np
import pytest

from oracle import oracle

SCALES = [0.1, 0.25, 0.35355, 0.5, 1.0]


def _4d(a):
    return np.ascontiguousarray(a.reshape((1, 1) + a.shape))


def test_scale_sweep_fp32(golden_dir):
    # test_scale_factor_fix.py:33-66 -- tolerance 1e-5 on max-abs
    g = np.load(golden_dir / "scale_sweep_fp32.npz")
    for s in (4, 8, 16, 32):
        q, k, v = (_4d(g[f"{n}_{s}"]) for n in "qkv")
        for sc in SCALES:
            o = oracle.sdpa_forward(q, k, v, scale=sc)[0, 0]
            assert np.abs(o - g[f"o_{s}_{sc}"]).max() < 1e-5
            assert np.abs(o - g[f"o64_{s}_{sc}"]).max() < 1e-6
        o = oracle.sdpa_forward(q, k, v)[0, 0]
        assert np.abs(o - g[f"o_{s}_default"]).max() < 1e-5


def test_known_answers(golden_dir):
    g = np.load(golden_dir / "known_answers.npz")
    ones = np.ones((1, 1, 4, 4), np.float32)
    o = oracle.sdpa_forward(ones, ones, ones, scale=0.5)
    assert np.abs(o - 1.0).max() < 1e-6 and np.abs(g["ones_o"] - 1.0).max() < 1e-5
    o = oracle.sdpa_forward(_4d(g["s1_q"]), _4d(g["s1_k"]), _4d(g["s1_v"]))
    assert np.abs(o[0, 0] - g["s1_v"]).max() == 0.0  # S=1 => O == V exactly
    assert np.abs(g["s1_o"] - g["s1_v"]).max() < 1e-6


@pytest.mark.parametrize("tag", ["1x1x64x64_fp32", "1x1x64x64_fp16", "1x1x64x64_bf16",
                                 "1x4x128x64_fp32", "1x4x128x64_fp16", "1x4x128x64_bf16",
                                 "1x1x512x128_bf16"])
def test_conftest_shapes(golden_dir, tag):
    g = np.load(golden_dir / "conftest_shapes.npz")
    q, k, v = (g[f"{n}_{tag}"] for n in "qkv")
    o = oracle.sdpa_forward(q, k, v)
    oc = oracle.sdpa_forward(q, k, v, causal=True)
    # both sides are fp64 math on identical rounded inputs, rounded once to fp32
    assert np.abs(o - g[f"o_{tag}"]).max() < 1e-6
    assert np.abs(oc - g[f"oc_{tag}"]).max() < 1e-6


def test_lcg_inputs(golden_dir):
    g = np.load(golden_dir / "lcg_inputs.npz")
    for name, shape in {"tiny": (1, 2, 4, 8), "small": (1, 4, 8, 16)}.items():
        n = int(np.prod(shape))
        q = oracle.lcg_uniform(n, 12345).reshape(shape)
        assert np.array_equal(q, g[f"q_{name}"])  # generator itself is pinned
        k, v = g[f"k_{name}"], g[f"v_{name}"]
        assert np.abs(oracle.sdpa_forward(q, k, v) - g[f"o_{name}"]).max() < 1e-6
        assert np.abs(oracle.sdpa_forward(q, k, v, causal=True) - g[f"oc_{name}"]).max() < 1e-6


def test_masks_and_ragged(golden_dir):
    g = np.load(golden_dir / "masks.npz")
    q, k, v = g["q"], g["k"], g["v"]
    assert np.abs(oracle.sdpa_forward(q, k, v) - g["o_dense"]).max() < 1e-6
    assert np.abs(oracle.sdpa_forward(q, k, v, causal=True) - g["o_causal"]).max() < 1e-6
    for name, mt in [("mask_bool_11qk", oracle.MASK_BOOL), ("mask_bool_b11k", oracle.MASK_BOOL),
                     ("mask_add_bhqk", oracle.MASK_ADDITIVE), ("mask_add_qk", oracle.MASK_ADDITIVE),
                     ("mask_add_hqk_fp16", oracle.MASK_ADDITIVE)]:
        o = oracle.sdpa_forward(q, k, v, mask=g[name], mask_type=mt)
        assert np.abs(o - g["o_" + name]).max() < 1e-6, name


def test_mask_strided_view(golden_dir):
    # element strides + size-1 broadcast (MFABridge.swift:193-196)
    g = np.load(golden_dir / "masks.npz")
    q, k, v = g["q"], g["k"], g["v"]
    big = np.zeros((g["mask_add_qk"].shape[0], 2 * g["mask_add_qk"].shape[1]), np.float32)
    big[:, ::2] = g["mask_add_qk"]
    o = oracle.sdpa_forward(q, k, v, mask=big[:, ::2], mask_type=oracle.MASK_ADDITIVE)
    assert np.abs(o - g["o_mask_add_qk"]).max() < 1e-6


@pytest.mark.parametrize("tag", ["dense", "causal"])
def test_lse_and_backward(golden_dir, tag):
    g = np.load(golden_dir / "backward_fp32.npz")
    q, k, v, do = (g[f"{n}_{tag}"] for n in ("q", "k", "v", "do"))
    o, lse = oracle.sdpa_forward(q, k, v, causal=(tag == "causal"), return_lse=True)
    assert np.abs(o - g[f"o_{tag}"]).max() < 1e-6
    assert np.abs(lse - g[f"lse_{tag}"]).max() < 1e-5
    dq, dk, dv, dvec = oracle.sdpa_backward(do, q, k, v, o, lse, causal=(tag == "causal"))
    for got, name in [(dq, "dq"), (dk, "dk"), (dv, "dv")]:
        ref = g[f"{name}_{tag}"]
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), name
    assert np.abs(dvec - (do * o).sum(-1)).max() < 1e-4


def test_strided_qkv(golden_dir):
    g = np.load(golden_dir / "masks.npz")
    q, k, v = g["q"], g["k"], g["v"]
    # [B,S,H,D] storage viewed as BHSD (test_stride_aware_attention.py) must equal contiguous
    qs = np.ascontiguousarray(q.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
    ks = np.ascontiguousarray(k.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
    vs = np.ascontiguousarray(v.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
    assert not qs.flags.c_contiguous
    assert np.array_equal(oracle.sdpa_forward(qs, ks, vs), oracle.sdpa_forward(q, k, v))


def test_quantiser_formula():
    # QuantizationTests.swift:72-128
    x = np.array([-1.0, -0.5, 0.0, 0.26, 0.5, 1.0, 0.004, -0.0039], np.float32)
    q, s = oracle.quantize_symmetric(x)
    assert np.isclose(s[0], 1.0 / 127.0)
    assert q.tolist() == [-127, -64, 0, 33, 64, 127, 1, 0]  # round half away from zero: 63.5 -> 64
    assert np.abs(oracle.dequantize(q, s) - x).max() <= s[0] / 2 + 1e-7
    q4, s4 = oracle.quantize_symmetric(x, bits=4)
    assert np.isclose(s4[0], 1.0 / 7.0) and q4.min() >= -8 and q4.max() <= 7
    packed = oracle.pack_int4(q4)
    assert packed[0] == ((int(q4[1]) + 8) << 4 | (int(q4[0]) + 8))  # even index in the low nibble
    assert np.array_equal(oracle.unpack_int4(packed, x.size), q4)
    z, sz = oracle.quantize_symmetric(np.zeros(5, np.float32))
    assert sz[0] == 1.0 and not z.any()  # absmax == 0 -> scale 1


def test_quantised_forward_error_budget():
    # docs/attic/PERFORMANCE_RESULTS.md:47-50 quotes INT8 ~0.1 %, INT4 ~2 % "typical" error; on
    # N(0,1) operands the formula itself (QuantizationTests.swift:72-128) gives ~1.5 % / ~30 %
    # relative L2, so the budgets here are the formula's, not the doc's.
    rng = np.random.default_rng(0)
    q, k, v = (rng.standard_normal((1, 2, 128, 64)).astype(np.float32) for _ in range(3))
    ref = oracle.sdpa_forward(q, k, v)
    for bits, budget in [(8, 5e-2), (4, 0.6)]:
        for mode in (0, 2):
            o, _ = oracle.quantized_forward(q, k, v, bits=bits, quant_mode=mode)
            rel = np.abs(o - ref).max() / np.abs(ref).max()
            assert rel < budget, (bits, mode, rel)


def test_rope_and_hadamard_oracle_identities():
    # rope: rotation then inverse rotation is the identity; norms of pairs preserved (MFABridge.swift:262-267)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 2, 5, 8)).astype(np.float32)
    ang = rng.uniform(0, 6.28, (5, 4)).astype(np.float32)
    cos, sin = np.repeat(np.cos(ang), 2, -1), np.repeat(np.sin(ang), 2, -1)
    y = oracle.rope_rotate(x, cos, sin)
    assert np.abs(oracle.rope_rotate(y, cos, sin, negate_sin=True) - x).max() < 1e-6
    assert np.allclose((y.reshape(-1, 2) ** 2).sum(-1), (x.reshape(-1, 2) ** 2).sum(-1), rtol=1e-5)
    assert np.allclose(y[0, 0, 0, 0], x[0, 0, 0, 0] * cos[0, 0] - x[0, 0, 0, 1] * sin[0, 0], atol=1e-6)
    # hadamard: H2 = [[1,1],[1,-1]]/sqrt2, involution
    h = oracle.hadamard(np.array([1.0, 0.0, 0.0, 0.0], np.float32), 4)
    assert np.allclose(h, [0.5, 0.5, 0.5, 0.5])
    v = rng.standard_normal(64).astype(np.float32)
    assert np.abs(oracle.hadamard(oracle.hadamard(v, 16).astype(np.float32), 16) - v).max() < 1e-6


def test_rope_oracle_matches_the_reference_eager_spec(golden_dir):
    """oracle.rope_rotate against fixtures computed by the reference's executable eager spec restated in torch
    (metal_sdpa_backend.cpp:1451-1468; tests/golden/gen_golden.py rotations): fp32 image within fp32 rounding of the
    spec's fp32 value, and -- rounded to the tensor's type -- the spec's own output bits for all but rounding ties."""
    g = np.load(golden_dir / "rope.npz")
    for name in ("fp32", "fp16", "bf16"):
        for lay in ("sd", "bsd"):
            tag = f"{name}_{lay}"
            x, cos, sin = g[f"x_{tag}"], g[f"cos_{tag}"], g[f"sin_{tag}"]
            y = oracle.rope_rotate(x, cos, sin)
            assert np.abs(y - g[f"y32_{tag}"]).max() < 2e-6 * max(1.0, np.abs(g[f"y32_{tag}"]).max()), tag
            if name == "fp16":
                got = y.astype(np.float16)
                want = g[f"y_{tag}"]
                assert (got != want).mean() < 1e-3 and np.abs(got.astype(np.float32) - want.astype(np.float32)).max() < 4e-3
            elif name == "bf16":
                got = oracle.f32_to_bf16_bits(y)
                assert (got != g[f"y_{tag}"]).mean() < 1e-3
            # inverse rotation (negate_sin): the gradient path of the reference's autograd wrapper
            back = oracle.rope_rotate(y.astype(np.float32), cos, sin, negate_sin=True)
            assert np.abs(back - oracle.to_f32(x)).max() < 1e-5 * max(1.0, np.abs(oracle.to_f32(x)).max())


def test_hadamard_oracle_matches_an_independent_sylvester_matrix(golden_dir):
    """oracle.hadamard (a butterfly) against y = H x / sqrt(N) with H from scipy.linalg.hadamard, computed as an explicit
    matrix product when the fixture was made: ordering (natural) and normalisation pinned by something the oracle did not compute"""
    g = np.load(golden_dir / "hadamard.npz")
    for n in (2, 16, 64, 256):
        y = oracle.hadamard(g[f"x_{n}"], n)
        assert np.abs(y - g[f"y_{n}"]).max() < 1e-6, n
