"""The decode form of the 128-row forward kernel (fa_fwd_16_kernel.h KS = 4; option "decode_ks"): at most 32 query rows per (batch, head), no mask, not causal.

The four waves of a workgroup all serve those rows -- wave w owns key quarter w of every 128-key tile -- and meet in LDS behind the sweep; the split-KV fold across
workgroups then takes one wave's slot per part.  (In the plain form three of four waves compute rows that do not exist and the launch waits for wave 0's chain through
whole 64-key tiles.)  Checked: oracle parity at the kernels' usual bounds for every instantiation, ragged key counts around the 128-key tile and its 32-key quarters,
forced split counts, strided operands, V outside fp16's range (each wave decides on its quarter's outputs, the workgroup sweeps again), LSE, bitwise repeatability,
graph replay, agreement with the plain form.  Reference behaviour matched: mfa_attention_forward / mfa_attention_encode_mtl with seq_len_q << seq_len_kv
(MFABridge.swift:2377-2543)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NORTH_STAR = 1.0e-3


def _oracle():
    from oracle import oracle
    return oracle


def npy(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.detach().cpu().contiguous().numpy()


def _rel(o, ref):
    return float(np.abs(o.float().cpu().numpy().astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("dtype,out,pv", [(torch.bfloat16, torch.float32, 1), (torch.bfloat16, torch.bfloat16, 1), (torch.bfloat16, torch.float32, 0), (torch.float16, torch.float32, 1)])
@pytest.mark.parametrize("Sq,Skv", [(1, 8192), (1, 130), (3, 1000), (17, 4097), (32, 64), (1, 33), (8, 127), (32, 2048)])
def test_decode_form_matches_the_oracle(D, dtype, out, pv, Sq, Skv):
    import umfa_torch
    torch.manual_seed(Sq * 7 + Skv + D)
    B, H = 2, 3
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=dtype)  # (6 items: far inside the plan's gate for the decode form)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=dtype) for _ in range(2))
    with umfa_torch.options(pv_fp16=pv):
        o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=out, return_lse=True)
        kern = umfa_torch.last_kernel()
        o2 = umfa_torch.attention_forward(q, k, v, out_dtype=out)
        with umfa_torch.options(decode_ks=2):
            o_plain = umfa_torch.attention_forward(q, k, v, out_dtype=out)
            assert "dec" not in umfa_torch.last_kernel()
    assert kern.endswith(",dec>"), kern
    assert torch.equal(o, o2)
    ref, rlse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), return_lse=True)
    tol = 4.0e-3 if (dtype == torch.bfloat16 and (pv == 0 or out == torch.bfloat16)) else NORTH_STAR
    assert _rel(o, ref) < tol, (kern, _rel(o, ref))
    assert np.abs(lse.cpu().numpy().reshape(rlse.shape) - rlse).max() < 2e-3
    assert _rel(o, o_plain.float().cpu().numpy().astype(np.float64)) < (1.6e-2 if out != torch.float32 else (4e-3 if (dtype == torch.bfloat16 and pv == 0) else 6e-4))  # (another deferred reference per key quarter: P round at other binade positions)


@pytest.mark.parametrize("parts", [0, 2, 5, 16])
@pytest.mark.parametrize("D", [64, 128])
def test_decode_form_with_forced_split_counts(parts, D):
    """the split-KV fold over workgroups takes the lead wave's slot of every part; parts with few (or no) 128-key tiles"""
    import umfa_torch
    torch.manual_seed(parts + D)
    B, H, Sq, Skv = 1, 4, 2, 3000
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    with umfa_torch.options(**({"force_split": parts} if parts else {})):
        o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
        assert umfa_torch.last_kernel().endswith(",dec>")
        o2 = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert torch.equal(o, o2)
    ref, rlse = _oracle().sdpa_forward(npy(q), npy(k), npy(v), return_lse=True)
    assert _rel(o, ref) < NORTH_STAR
    assert np.abs(lse.cpu().numpy().reshape(rlse.shape) - rlse).max() < 2e-3


@pytest.mark.parametrize("kind", ["outlier", "tiny", "huge", "row_scaled"])
@pytest.mark.parametrize("D", [64, 128])
def test_decode_form_over_the_range_of_v(kind, D):
    """the converting kernel's range check in the decode form: a wave looks at ITS key quarter's outputs, the workgroup sweeps again with the slab's shift"""
    import umfa_torch
    torch.manual_seed(11)
    B, H, Sq, Skv = 2, 2, 1, 5000
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    if kind == "outlier":
        v[0, 1, 4100, 7] = 3.0e8
    elif kind == "tiny":
        v = (v.float() * 1e-7).to(torch.bfloat16)
    elif kind == "huge":
        v = (v.float() * 1e20).to(torch.bfloat16)
    else:
        v[:, :, Skv // 2:] = (v[:, :, Skv // 2:].float() * 4096.0).to(torch.bfloat16)
    o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    assert umfa_torch.last_kernel().endswith(",pv16,dec>")
    assert torch.isfinite(o).all()
    ref = _oracle().sdpa_forward(npy(q), npy(k), npy(v))
    for b in range(B):
        for h in range(H):
            d = np.abs(ref[b, h]).max()
            assert np.abs(o[b, h].cpu().numpy() - ref[b, h]).max() / d < (1.5e-3 if kind == "row_scaled" else NORTH_STAR), (kind, b, h)
    assert torch.equal(o, umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32))


def test_decode_form_strided_operands_and_graph_replay():
    import umfa_torch
    torch.manual_seed(4)
    B, H, Sq, Skv, D = 2, 8, 1, 2500, 128
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)  # a BSHD cache viewed as BHSD
    v = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)
    out = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    umfa_torch.attention_forward(q, k, v, out=out)
    assert umfa_torch.last_kernel().endswith(",dec>")
    assert _rel(out, _oracle().sdpa_forward(npy(q), npy(k.contiguous()), npy(v.contiguous()))) < NORTH_STAR
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        umfa_torch.attention_forward(q, k, v, out=out)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            umfa_torch.attention_forward(q, k, v, out=out)
        for rep in range(3):
            if rep:
                q.copy_(torch.randn_like(q)); v.copy_(torch.randn(B, Skv, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2) * (1.0 if rep == 1 else 1e4))
            g.replay()
            side.synchronize()
            assert _rel(out, _oracle().sdpa_forward(npy(q), npy(k.contiguous()), npy(v.contiguous()))) < NORTH_STAR, rep
    torch.cuda.current_stream().wait_stream(side)
