"""GPU parity of the two 'next' rows built so far: rotary rotation and Hadamard rotation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _oracle():
    from oracle import oracle
    return oracle


def _tables(S, D, B=None, seed=0):
    # pair-duplicated fp32 tables as the reference builds them (metal_sdpa_backend.cpp:1451-1468)
    g = torch.Generator().manual_seed(seed)
    shape = (S, D // 2) if B is None else (B, S, D // 2)
    ang = torch.rand(shape, generator=g) * 6.283
    return ang.cos().repeat_interleave(2, -1).contiguous(), ang.sin().repeat_interleave(2, -1).contiguous()


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 40, 64), (1, 4, 17, 128), (1, 2, 9, 6)])
def test_rope_matches_oracle_and_inverts(dt, shape):
    import umfa_torch
    orc = _oracle()
    B, H, S, D = shape
    torch.manual_seed(1)
    x = torch.randn(shape, dtype=dt)
    for batched in (False, True):
        cos, sin = _tables(S, D, B if batched else None)
        xb = x.numpy() if dt != torch.bfloat16 else x.view(torch.int16).numpy().view(np.uint16)
        ref = orc.rope_rotate(xb, cos.numpy(), sin.numpy())
        y = umfa_torch.rope_rotate(x.cuda(), cos.cuda(), sin.cuda())
        assert y.dtype == dt and y.shape == x.shape
        tol = {torch.float32: 1e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dt]  # one rounding to the type
        assert (y.float().cpu() - torch.from_numpy(ref)).abs().max() < tol * max(1.0, float(np.abs(ref).max()))
        back = umfa_torch.rope_rotate(y, cos.cuda(), sin.cuda(), negate_sin=True)  # inverse rotation
        assert (back.float().cpu() - x.float()).abs().max() < 3 * tol * max(1.0, float(x.float().abs().max()))


def test_rope_strided_source_equals_contiguous():
    import umfa_torch
    base = torch.randn(2, 50, 4, 64, device="cuda", dtype=torch.bfloat16)  # [B,S,H,D] storage
    x = base.permute(0, 2, 1, 3)
    cos, sin = _tables(50, 64)
    a = umfa_torch.rope_rotate(x, cos.cuda(), sin.cuda())
    b = umfa_torch.rope_rotate(x.contiguous(), cos.cuda(), sin.cuda())
    assert torch.equal(a, b)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("block", [2, 64, 256, 4096])
def test_hadamard_matches_oracle_and_is_involution(dt, block):
    import umfa_torch
    orc = _oracle()
    torch.manual_seed(2)
    x = torch.randn(3 * 4096, dtype=dt)
    ref = orc.hadamard(x.numpy(), block)
    y = umfa_torch.hadamard_rotate(x.cuda().clone(), block)
    tol = 2e-5 if dt == torch.float32 else 4e-3
    assert np.abs(y.float().cpu().numpy() - ref).max() < tol * max(1.0, float(np.abs(ref).max()))
    z = umfa_torch.hadamard_rotate(y.clone(), block)  # H H = I (AGENTS.md:161-170)
    assert (z.float().cpu() - x.float()).abs().max() < 3 * tol * max(1.0, float(x.float().abs().max()))
    # energy preserving (orthonormal): outlier smoothing without changing norms
    assert abs(float(y.float().norm()) / float(x.float().norm()) - 1.0) < 2e-3
    with pytest.raises(RuntimeError):
        umfa_torch.hadamard_rotate(torch.zeros(10, device="cuda"), 4)


@pytest.mark.parametrize("name", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("lay", ["sd", "bsd"])
def test_rope_kernel_matches_the_reference_eager_spec_fixture(golden_dir, name, lay):
    """the HIP rotation against tests/golden/rope.npz (the reference's eager spec, metal_sdpa_backend.cpp:1451-1468)"""
    import umfa_torch
    g = np.load(golden_dir / "rope.npz")
    tag = f"{name}_{lay}"
    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[name]
    xb = g[f"x_{tag}"]
    x = (torch.from_numpy(xb.view(np.int16)).view(torch.bfloat16) if name == "bf16" else torch.from_numpy(xb)).cuda()
    cos, sin = torch.from_numpy(g[f"cos_{tag}"]).cuda(), torch.from_numpy(g[f"sin_{tag}"]).cuda()
    y = umfa_torch.rope_rotate(x, cos, sin)
    assert y.dtype == dt
    wb = g[f"y_{tag}"]
    want = torch.from_numpy(wb.view(np.int16)).view(torch.bfloat16) if name == "bf16" else torch.from_numpy(wb)
    diff = (y.float().cpu() - want.float()).abs()
    tol = {"fp32": 2e-6, "fp16": 2e-3, "bf16": 1.6e-2}[name]  # one ulp of the type at the largest value (ties may round either way)
    assert float(diff.max()) <= tol * max(1.0, float(want.float().abs().max()))
    if name != "fp32":  # (fp32: the kernel's pinned fma(x0, c, -(x1 * s)) and the spec's two rounded products differ by an ulp)
        assert float((diff > 0).float().mean()) < 2e-3  # almost every element is the spec's exact bits


@pytest.mark.parametrize("n", [16, 64, 256])
def test_hadamard_kernel_matches_the_sylvester_fixture(golden_dir, n):
    import umfa_torch
    g = np.load(golden_dir / "hadamard.npz")
    y = umfa_torch.hadamard_rotate(torch.from_numpy(g[f"x_{n}"]).cuda(), n)
    assert np.abs(y.cpu().numpy() - g[f"y_{n}"]).max() < 2e-5
