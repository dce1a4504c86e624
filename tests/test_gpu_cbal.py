"""Balanced causal pairs on the 128-row forward kernel (fa_fwd_16_kernel.h CBAL; option "cbal").

A causal launch that is resident at once ends with its longest q-block alone on its CU.  The paired schedule deals the q-blocks
(i, nqb - 1 - i) of a head to two workgroups of equal length: part B sweeps the long block's tail, publishes (O^T, m, l) mid-sweep
and goes on with the short block; part A sweeps the long block's head and folds B's part in.  Checked here: oracle parity at the
kernels' usual bounds for every instantiation (bf16 pv16 in-kernel conversion and cast pre-pass, bf16 P V, fp16; fp32 and operand-type
O; LSE), every cut position, ragged and unequal Sq / Skv, V outside fp16's range placed so that EACH part has to decide on its own
second sweep, bitwise repeatability, hipGraph replay with changing data, and the plan's gate.  Reference behaviour matched: causal
SDPA through mfa_attention_encode_mtl (MFABridge.swift:2377-2543; causal handling MFABridge.swift:2205-2248)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NORTH_STAR = 1.0e-3


def _oracle():
    from oracle import oracle
    return oracle


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _ref(q, k, v, causal=True, lse=False):
    conv = bits if q.dtype == torch.bfloat16 else (lambda t: t.detach().cpu().numpy())
    return _oracle().sdpa_forward(conv(q), conv(k), conv(v), causal=causal, return_lse=lse)


def _rel(o, ref):
    return float(np.abs(o.float().cpu().numpy().astype(np.float64) - ref).max() / np.abs(ref).max())


def _run(q, k, v, delta=-1, **kw):
    import umfa_torch
    with umfa_torch.options(cbal=1, cbal_delta=delta, no_w64=1):
        o = umfa_torch.attention_forward(q, k, v, causal=True, **kw)
        kern = umfa_torch.last_kernel()
    return o, kern


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("dtype,out,pv", [(torch.bfloat16, torch.float32, 1), (torch.bfloat16, torch.bfloat16, 1), (torch.bfloat16, torch.float32, 0),
                                          (torch.float16, torch.float32, 1), (torch.float16, torch.float16, 1)])
@pytest.mark.parametrize("S", [256, 512, 1024])
def test_paired_schedule_matches_the_oracle(D, dtype, out, pv, S):
    import umfa_torch
    torch.manual_seed(S + D)
    q, k, v = (torch.randn(2, 3, S, D, device="cuda", dtype=dtype) for _ in range(3))
    with umfa_torch.options(pv_fp16=pv):
        (o, lse), kern = _run(q, k, v, out_dtype=out, return_lse=True)
        with umfa_torch.options(cbal=2, no_w64=1):
            o_plain = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=out)
    assert kern.startswith("fa_fwd16<"), kern
    ref, rlse = _ref(q, k, v, lse=True)
    tol = 4.0e-3 if (dtype == torch.bfloat16 and (pv == 0 or out == torch.bfloat16)) else (1.0e-3 if out == torch.float32 else 2.0e-3)
    assert _rel(o, ref) < tol, (kern, _rel(o, ref))
    assert np.abs(lse.cpu().numpy().reshape(rlse.shape) - rlse).max() < 2e-3
    # the unpaired schedule computes the same tiles: the two agree far inside the tolerance (another order of the row sums, one fold)
    # (bf16 P: a part that starts above tile 0 has another deferred reference, so its P round at other binade positions -- the format's own ulp)
    close = 1.6e-2 if out != torch.float32 else (4e-3 if (dtype == torch.bfloat16 and pv == 0) else 3e-4)
    assert _rel(o, o_plain.float().cpu().numpy().astype(np.float64)) < close


@pytest.mark.parametrize("delta", [0, 1, 2, 3, 5, 16])
@pytest.mark.parametrize("D", [64, 128])
def test_every_cut_position(delta, D):
    """cbal_delta moves the cut between the parts: down to one tile for part A, and cuts ABOVE the diagonal of the long block's first rows
    (part B then starts with rows that see none of its keys)"""
    torch.manual_seed(3)
    q, k, v = (torch.randn(1, 4, 768, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o, kern = _run(q, k, v, delta=delta, out_dtype=torch.float32)
    assert _rel(o, _ref(q, k, v)) < NORTH_STAR, (delta, D, kern)


@pytest.mark.parametrize("Sq,Skv", [(512, 640), (512, 300), (500, 500), (384, 1024), (1024, 64), (256, 130), (384, 384), (640, 640), (1152, 1152), (1100, 1100), (640, 200)])
@pytest.mark.parametrize("D", [64, 128])
def test_ragged_and_unequal_lengths(Sq, Skv, D):
    """causal is top-left aligned (key <= row): Skv < Sq caps the long blocks (pairs that need no cut stay whole), ragged last tiles and rows; an ODD
    number of q-blocks leaves the middle one -- as long as half a pair -- whole, behind the pairs in the grid"""
    torch.manual_seed(Sq + Skv)
    q = torch.randn(2, 2, Sq, D, device="cuda", dtype=torch.bfloat16)
    k, v = (torch.randn(2, 2, Skv, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    for delta in (0, 2):
        o, kern = _run(q, k, v, delta=delta, out_dtype=torch.float32)
        assert torch.isfinite(o).all()
        assert _rel(o, _ref(q, k, v)) < NORTH_STAR, (Sq, Skv, D, delta, kern)


@pytest.mark.parametrize("key,D", [(1000, 64), (40, 64), (600, 64), (1000, 128), (40, 128)])
def test_each_part_decides_its_own_second_sweep(key, D):
    """the converting kernel (pv16 = 1): a V value beyond fp16's range makes the workgroups that stage its tile sweep again with the slab's
    shift.  Key 1000 sits in the long blocks' tails (parts B decide AT THE SWITCH, before they publish), key 40 in tile 0 (every part A and
    every short block), key 600 in between (both kinds)"""
    import umfa_torch
    torch.manual_seed(5)
    q, k, v = (torch.randn(1, 2, 1024, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    v[0, 1, key, 3] = -7.0e9
    o, kern = _run(q, k, v, out_dtype=torch.float32)
    assert "pv16" in kern, kern
    assert torch.isfinite(o).all()
    ref = _ref(q, k, v)
    for h in range(2):
        d = np.abs(ref[0, h]).max()
        assert np.abs(o[0, h].cpu().numpy() - ref[0, h]).max() / d < NORTH_STAR, (key, D, h)
    o2, _ = _run(q, k, v, out_dtype=torch.float32)
    assert torch.equal(o, o2)
    # tiny V: every output below 2^-11 -- the other trigger of the second sweep
    vt = (torch.randn(1, 2, 1024, D, device="cuda") * 1e-7).to(torch.bfloat16)
    ot, _ = _run(q, k, vt, out_dtype=torch.float32)
    reft = _ref(q, k, vt)
    assert np.abs(ot.cpu().numpy() - reft).max() / np.abs(reft).max() < NORTH_STAR


def test_second_sweep_publishes_once():
    """a part B whose tail was fine (published, flag raised) and whose SHORT block then needs the second sweep must not publish again: the
    flag would stay up behind the launch and the next launch's part A would fold a slot that is still being written (found by
    tools/lab/value_fuzz.py run_cbal_case: launches after such a one were not repeatable)"""
    torch.manual_seed(21)
    q, k, v = (torch.randn(2, 3, 1024, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    vx = v.clone()
    vx[:, :, 40, 5] = 3.0e8  # tile 0: every short block and every part A see it, no long block's tail does
    first, _ = _run(q, k, v, out_dtype=torch.float32)
    for _ in range(6):
        ox, _ = _run(q, k, vx, out_dtype=torch.float32)
        assert torch.isfinite(ox).all()
        again, _ = _run(q, k, v, out_dtype=torch.float32)
        assert torch.equal(first, again)
    assert _rel(first, _ref(q, k, v)) < NORTH_STAR


def test_bitwise_repeatable_and_graph_replay_with_changing_data():
    """the fold is one fixed order (A's registers, then B's slot): launches repeat bit for bit; the pairs' flags are left zero, so a captured
    graph replays with other data and gives that data's result"""
    torch.manual_seed(9)
    q, k, v = (torch.randn(2, 4, 1024, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o1, kern = _run(q, k, v, out_dtype=torch.float32)
    for _ in range(5):
        o2, _ = _run(q, k, v, out_dtype=torch.float32)
        assert torch.equal(o1, o2)
    import umfa_torch
    out = torch.empty_like(o1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with umfa_torch.options(cbal=1, no_w64=1):
        with torch.cuda.stream(side):
            umfa_torch.attention_forward(q, k, v, causal=True, out=out)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                umfa_torch.attention_forward(q, k, v, causal=True, out=out)
            for rep in range(4):
                if rep:
                    q.copy_(torch.randn_like(q)); v.copy_(torch.randn_like(v) * (3.0 if rep == 2 else 1.0))
                g.replay()
                side.synchronize()
                assert _rel(out, _ref(q, k, v)) < NORTH_STAR, rep
    torch.cuda.current_stream().wait_stream(side)


def test_the_plans_gate():
    """default options: the paired schedule where the plan expects a gain, the unpaired one elsewhere; both inside the tolerance.  (The
    kernel name is the same -- the schedule is a property of the launch; the gate shows in the timing records, profiles/r6/cbal_matrix.jsonl,
    and here through the agreement of default and forced results bit for bit)"""
    import umfa_torch
    torch.manual_seed(2)
    for (B, H, S, D, paired) in [(1, 8, 2048, 128, True), (1, 2, 1024, 128, True), (4, 16, 1024, 64, False), (1, 8, 512, 128, False), (1, 4, 2048, 64, True)]:
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        with umfa_torch.options(no_w64=1):
            o = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
            with umfa_torch.options(cbal=1):
                o_on = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
            with umfa_torch.options(cbal=2):
                o_off = umfa_torch.attention_forward(q, k, v, causal=True, out_dtype=torch.float32)
        assert torch.equal(o, o_on if paired else o_off), (B, H, S, D)
        assert not torch.equal(o_on, o_off)  # (the two schedules differ in the last bits: the comparison above means something)
