"""umfa_rope_attention_forward_stream: RoPE + SDPA in one call (K rotated once into the stream's workspace, Q rotated inside
the 256-row attention kernel).  The reference's sequence is rotate(q), rotate(k), attend
(metal_sdpa_backend.cpp:1472-1641); the fused entry must give the SAME BITS as that sequence on every kernel it can land
on (in-register rotation on fa_fwd16_w64, pre-pass on the 128-row / exact kernels), and the oracle's
rope_rotate -> sdpa_forward within the path's usual bounds."""
import numpy as np
import pytest
from tolerances import fam

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _tables(S, D, B=None, seed=5):
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand((S, D // 2) if B is None else (B, S, D // 2), generator=g) * 6.283
    return ang.cos().repeat_interleave(2, -1).cuda(), ang.sin().repeat_interleave(2, -1).cuda()


def _raw(t):
    t = t.cpu().contiguous()
    return t.view(torch.int32 if t.dtype == torch.float32 else torch.int16).numpy()


CASES = [  # (B, H, S, D), dtype, causal, batched tables, force the 256-row kernel
    ((1, 24, 1024, 128), torch.bfloat16, False, False, True),
    ((2, 3, 512, 128), torch.bfloat16, True, True, True),
    ((1, 4, 1280, 128), torch.float16, False, False, True),   # ragged last 256-row block
    ((2, 2, 768, 128), torch.float16, True, True, True),
    ((1, 48, 2048, 128), torch.bfloat16, True, False, False),  # large enough for the w64 plan without forcing
    ((2, 4, 192, 64), torch.bfloat16, True, False, False),    # 128-row kernel: Q through the pre-pass
    ((1, 2, 320, 128), torch.float16, False, True, False),
    ((1, 2, 96, 80), torch.float32, False, False, False),     # exact kernel
    ((1, 3, 200, 256), torch.bfloat16, True, True, False),
]


@pytest.mark.parametrize("shape,dt,causal,batched,force", CASES)
def test_fused_equals_rotate_then_attend(shape, dt, causal, batched, force, umfa_opts):
    import umfa_torch
    from umfa_torch import ops
    if force:
        umfa_opts(force_w64=1)
    B, H, S, D = shape
    torch.manual_seed(11)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
    cos, sin = _tables(S, D, B if batched else None)
    out, lse = ops.rope_attention_forward(q, k, v, cos, sin, causal=causal, return_lse=True)
    name = umfa_torch.last_kernel()
    if force or shape == (1, 48, 2048, 128):
        assert name.startswith("fa_fwd16_w64<") and name.endswith(",rope>"), name
    ref, lse_ref = ops.attention_forward(ops.rope_rotate(q, cos, sin), ops.rope_rotate(k, cos, sin), v, causal=causal,
                                         return_lse=True)
    assert np.array_equal(_raw(out), _raw(ref)), f"{name}: fused O differs from rotate-then-attend"
    assert np.array_equal(_raw(lse), _raw(lse_ref))


def test_fused_strided_operands_and_routing(umfa_opts):
    """BSHD-permuted views (contiguous last dim only) through the public routing function."""
    import umfa_torch
    from umfa_torch import ops
    umfa_opts(force_w64=1)
    B, H, S, D = 1, 4, 512, 128
    torch.manual_seed(12)
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3) for _ in range(3))
    cos, sin = _tables(S, D)
    umfa_torch.reset_dispatch_stats()
    out = umfa_torch.rope_scaled_dot_product_attention(q, k, v, cos, sin)
    assert fam(umfa_torch.last_kernel()) == "fa_fwd16_w64<bf16,128,rope>"
    st = umfa_torch.get_dispatch_stats()
    assert st["total"] == 1 and st["rope_instream"] == 1 and st["fp32_instream"] == 1
    ref = ops.attention_forward(ops.rope_rotate(q, cos, sin), ops.rope_rotate(k, cos, sin), v)
    assert np.array_equal(_raw(out), _raw(ref))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_fused_vs_oracle(dt, umfa_opts):
    import umfa_torch  # noqa: F401
    from umfa_torch import ops
    from oracle import oracle
    from tolerances import check_forward
    umfa_opts(force_w64=1)
    B, H, S, D = 1, 2, 512, 128
    torch.manual_seed(13)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
    cos, sin = _tables(S, D)
    out = ops.rope_attention_forward(q, k, v, cos, sin, causal=True)
    c, s_ = cos.cpu().numpy(), sin.cpu().numpy()
    if dt == torch.bfloat16:  # the oracle's bf16 operands are uint16 bit images
        qn, kn, vn = (t.cpu().view(torch.int16).numpy().view(np.uint16) for t in (q, k, v))
        qr, kr = oracle.f32_to_bf16_bits(oracle.rope_rotate(qn, c, s_)), oracle.f32_to_bf16_bits(oracle.rope_rotate(kn, c, s_))
    else:
        qn, kn, vn = (t.cpu().numpy() for t in (q, k, v))
        qr, kr = oracle.rope_rotate(qn, c, s_).astype(np.float16), oracle.rope_rotate(kn, c, s_).astype(np.float16)
    ref = oracle.sdpa_forward(qr, kr, vn, causal=True)
    check_forward(out.float().cpu().numpy(), ref, dt, "fa_fwd16_w64<rope>", "rope_fused", out_dt=dt)


def test_fused_rejects_bad_arguments():
    import umfa_torch  # noqa: F401
    from umfa_torch import ops
    q = torch.randn(1, 2, 64, 63, device="cuda", dtype=torch.bfloat16)     # odd head_dim
    cos = torch.zeros(64, 63, device="cuda"); sin = torch.zeros(64, 63, device="cuda")
    from umfa._ffi import MFAError
    with pytest.raises(MFAError):
        ops.rope_attention_forward(q, q, q, cos, sin)
    q = torch.randn(1, 2, 64, 64, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 2, 96, 64, device="cuda", dtype=torch.bfloat16)     # Sq != Skv
    cos = torch.zeros(64, 64, device="cuda")
    with pytest.raises(MFAError):
        ops.rope_attention_forward(q, k, k, cos, cos)
