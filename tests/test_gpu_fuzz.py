"""GPU: seeded random sweep over shapes, head dims, dtypes, causal, masks and strided views -- forward and
backward through the C ABI (umfa_torch.attention_forward / the autograd Function behind
scaled_dot_product_attention) against a plain fp32 PyTorch restatement of the same op evaluated on the GPU
(softmax(scale * Q K^T + mask) V, top-left causal alignment like torch's is_causal).  The oracle covers the small
cases bit-for-bit elsewhere; this file is about dispatch: every shape lands on SOME kernel (w64, 128-row, exact,
split-KV, ragged tails) and all of them must agree with the same reference to the tolerance of their precision.
Tolerances: rel = max|O - O_ref| / max|O_ref| <= 6e-3 bf16, 2e-3 fp16, 2e-5 fp32."""
import random

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = {torch.bfloat16: 6e-3, torch.float16: 2e-3, torch.float32: 2e-5}
FUZZ_SCALE = 1.5  # random masks / ragged tiny shapes: measured worst case over the 96 seeds, see profiles/r2/parity_measured.json
GTOL = {torch.bfloat16: 3e-2, torch.float16: 8e-3, torch.float32: 1e-4}  # gradients: P and dS rounded to the input type


def _ref(q, k, v, scale, causal, mask):
    if q.dtype in (torch.float32, torch.float64):  # fp32 operands: the restatement itself in fp64
        q, k, v = q.double(), k.double(), v.double()
        s = torch.matmul(q, k.transpose(-1, -2)) * scale
        Sq, Skv = s.shape[-2:]
        if causal:
            s = s.masked_fill(~torch.ones(Sq, Skv, dtype=torch.bool, device=q.device).tril(), float("-inf"))
        if mask is not None:
            s = s.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else s + mask.double()
        return torch.matmul(torch.softmax(s, dim=-1), v)
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    Sq, Skv = s.shape[-2:]
    if causal:
        keep = torch.ones(Sq, Skv, dtype=torch.bool, device=q.device).tril()
        s = s.masked_fill(~keep, float("-inf"))
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else s + mask.float()
    return torch.matmul(torch.softmax(s, dim=-1), v.float())


def _case(seed):
    rng = random.Random(seed)
    dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
    D = rng.choice([32, 64, 64, 80, 128, 128, 128, 256] if dt != torch.float32 else [32, 64, 96, 128])
    B, H = rng.choice([1, 1, 2, 3]), rng.choice([1, 2, 3, 6])
    big = rng.random() < 0.4
    Sq = rng.choice([1, 7, 64, 100, 128, 255, 256, 257, 333]) if not big else rng.choice([512, 640, 1000, 1024, 1280, 1531, 2048])
    Skv = Sq if rng.random() < 0.6 else rng.choice([1, 17, 63, 64, 65, 200, 256, 511, 777, 1024, 1100])
    if dt == torch.float32:
        Sq, Skv = min(Sq, 640), min(Skv, 640)
    causal = rng.random() < 0.4
    mk = rng.choice([None, None, None, "bool", "float", "bcast", "strided", "window"]) if not causal else None
    strided = rng.random() < 0.3
    return dt, B, H, Sq, Skv, D, causal, mk, strided


def _tensors(seed, dt, B, H, Sq, Skv, D, strided):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if strided:  # BSHD storage viewed as BHSD: head stride D, row stride H*D (the layout FLUX-style callers hand over)
        q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt, generator=g).transpose(1, 2)
        k = torch.randn(B, Skv, H, D, device="cuda", dtype=dt, generator=g).transpose(1, 2)
        v = torch.randn(B, Skv, H, D, device="cuda", dtype=dt, generator=g).transpose(1, 2)
    else:
        q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt, generator=g)
        k = torch.randn(B, H, Skv, D, device="cuda", dtype=dt, generator=g)
        v = torch.randn(B, H, Skv, D, device="cuda", dtype=dt, generator=g)
    return q, k, v, g


def _mask(kind, B, H, Sq, Skv, g):
    if kind is None:
        return None
    if kind == "bool":
        m = torch.rand(B, H, Sq, Skv, device="cuda", generator=g) < 0.8
        m[..., 0] = True  # no fully masked row
        return m
    if kind == "float":
        return torch.randn(B, 1, Sq, Skv, device="cuda", generator=g)
    if kind == "strided":  # every second column of a wider mask: key stride 2 (no vector reads), per-head
        m = (torch.rand(B, H, Sq, 2 * Skv, device="cuda", generator=g) < 0.6)[..., ::2]
        m[..., 0] = True
        return m
    if kind == "window":   # band mask: most tiles fully masked or fully open (the tile-flag fast paths)
        i = torch.arange(Sq, device="cuda")[:, None]
        j = torch.arange(Skv, device="cuda")[None, :]
        return ((i * Skv // max(Sq, 1) - j).abs() <= max(8, Skv // 6))[None, None]
    m = torch.rand(1, 1, 1, Skv, device="cuda", generator=g) < 0.7  # key-padding style broadcast mask
    m[..., 0] = True
    return m


@pytest.mark.parametrize("seed", range(96))
def test_forward_random_case(seed):
    import umfa_torch
    dt, B, H, Sq, Skv, D, causal, mk, strided = _case(seed)
    q, k, v, g = _tensors(seed, dt, B, H, Sq, Skv, D, strided)
    mask = _mask(mk, B, H, Sq, Skv, g)
    scale = D ** -0.5
    ref = _ref(q, k, v, scale, causal, mask)
    out, lse = umfa_torch.attention_forward(q, k, v, causal=causal, mask=mask, out_dtype=torch.float32, return_lse=True)
    what = (seed, dt, B, H, Sq, Skv, D, causal, mk, strided, umfa_torch.last_kernel())
    assert torch.isfinite(out).all(), what
    rel = ((out.to(ref.dtype) - ref).abs().max() / ref.abs().max()).item()
    if dt == torch.float32:
        assert rel < TOL[dt], (rel, what)
    else:  # measured bounds (tests/tolerances.py); short key ranges and sparse masks put more weight on single keys
        from tolerances import check_forward
        check_forward(out.double().cpu().numpy(), ref.double().cpu().numpy(), dt, umfa_torch.last_kernel(), f"fuzz{seed}",
                      scale_max=FUZZ_SCALE)
    # log-sum-exp (natural log) of the scaled, masked scores
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(~torch.ones(Sq, Skv, dtype=torch.bool, device="cuda").tril(), float("-inf"))
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else s + mask.float()
    assert (lse.view(B, H, Sq) - torch.logsumexp(s, dim=-1)).abs().max().item() < 2e-2 * (1 if dt != torch.float32 else 0.01), what
    # the fused cast-back epilogue agrees with the fp32 output rounded once
    if dt != torch.float32:
        o2 = umfa_torch.attention_forward(q, k, v, causal=causal, mask=mask)
        assert o2.dtype == dt and ((o2.float() - out).abs().max() / out.abs().max()).item() <= (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * 1.01, what  # one rounding of O


@pytest.mark.parametrize("seed", range(100, 164))
def test_backward_random_case(seed):
    import umfa_torch
    dt, B, H, Sq, Skv, D, causal, mk, strided = _case(seed)
    mk = None  # the backward ABI (mfa_attention_backward) takes no mask
    Sq, Skv = min(Sq, 1024), min(Skv, 1024)
    q, k, v, g = _tensors(seed, dt, B, H, Sq, Skv, D, False)
    do = torch.randn(B, H, Sq, D, device="cuda", dtype=dt, generator=g)
    ref_dt = torch.float64 if dt == torch.float32 else torch.float32
    qr, kr, vr = (t.detach().to(ref_dt).requires_grad_(True) for t in (q, k, v))
    _ref(qr, kr, vr, D ** -0.5, causal, None).backward(do.to(ref_dt))
    qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    out = umfa_torch.scaled_dot_product_attention(qg, kg, vg, is_causal=causal)
    out.backward(do)
    what = (seed, dt, B, H, Sq, Skv, D, causal, umfa_torch.last_kernel())
    for got, ref, name in [(qg.grad, qr.grad, "dq"), (kg.grad, kr.grad, "dk"), (vg.grad, vr.grad, "dv")]:
        assert got is not None and torch.isfinite(got).all(), (name, what)
        # Skv = 1 makes dQ exactly 0 in exact arithmetic (dS = P (dP - D) with P = 1, D = dP): the floor of the
        # denominator keeps the metric meaningful there (operands are N(0,1), ordinary gradients are O(0.1 .. 1))
        rel = ((got.to(ref.dtype) - ref).abs().max() / ref.abs().max().clamp_min(0.05)).item()
        if Skv == 1 and name != "dv":
            # ... and what is measured there is the cancellation noise of dP - D (both O(10), D from the rounded O the
            # caller hands back), summed over the queries: bounded absolutely, not against an exact zero
            assert (got.to(ref.dtype) - ref).abs().max().item() < 1e-6 * Sq + GTOL[dt] * 0.05, (name, what)
            continue
        assert rel < GTOL[dt], (name, rel, what)
