"""Asynchronous entries under concurrency and graph capture (csrc/runtime_internal.h "scratch"):
scratch is per (device, stream) and per capture, grow-only (csrc/runtime_internal.h)."""
import threading

import numpy as np
import pytest
from tolerances import fam

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True)
def _force_w64(umfa_opts):
    umfa_opts(force_w64=1)  # small grids: every item of fa_fwd16_w64 is cut into parts and folded


def _inputs(seed, B=1, H=5, Sq=768, Skv=448):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return tuple(torch.randn(B, H, s, 128, device="cuda", dtype=torch.bfloat16, generator=g) for s in (Sq, Skv, Skv))


def test_two_streams_with_cut_items_equal_serial():
    """Two streams launch forwards whose items are cut (tickets + partials in scratch) at the same time; each stream has
    its own ticket words and partial slots, so every result is bit-equal to the serial run."""
    import umfa_torch
    cases = [_inputs(s, H=6, Sq=4096, Skv=4096) for s in range(4)]
    serial = [umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32) for q, k, v in cases]
    assert umfa_torch.last_kernel().startswith("fa_fwd16_w64")
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = {}
    for rep in range(6):  # interleaved launches, no synchronisation in between: the two streams overlap on the GPU
        for i, (q, k, v) in enumerate(cases):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                outs[(rep, i)] = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
    torch.cuda.synchronize()
    for (rep, i), o in outs.items():
        assert torch.equal(o, serial[i]), (rep, i)


def test_two_host_threads_two_streams():
    import umfa_torch
    cases = [_inputs(10 + s) for s in range(2)]
    serial = [umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32) for q, k, v in cases]
    torch.cuda.synchronize()
    res, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(50):
                    o = umfa_torch.attention_forward(*cases[i], out_dtype=torch.float32)
            st.synchronize()
            res[i] = o
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert torch.equal(res[0], serial[0]) and torch.equal(res[1], serial[1])


def test_graph_survives_scratch_growth_and_capture_never_allocates():
    """A captured graph keeps replaying correctly after later, larger calls on the same stream (the capture took the
    warmed-up pool over as its own; eager calls afterwards fill a fresh one); nothing is allocated inside a capture: a shape whose
    kernel would need NEW scratch there runs on the kernel that needs none (round 5: the 128-row kernel with V converted in the
    kernel -- until then such a call was refused with MFA_ERROR_MEMORY_ALLOCATION); a call that cannot run without new scratch
    (a split-KV plan larger than any the pool has seen) still is."""
    import umfa_torch
    from umfa._ffi import MFAError
    q, k, v = _inputs(1)
    mask = torch.ones(1, 1, 768, 448, dtype=torch.bool, device="cuda").tril(100)
    out_a = torch.empty(1, 5, 768, 128, device="cuda", dtype=torch.float32)
    out_m = torch.empty_like(out_a)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        # warm-up on the capture stream: its pool gets the w64 partials, the mask-flag buffer, the split scratch
        umfa_torch.attention_forward(q, k, v, out=out_a)
        umfa_torch.attention_forward(q, k, v, mask=mask, out=out_m)
    side.synchronize()
    ref_a, ref_m = out_a.clone(), out_m.clone()
    big = _inputs(2, H=24, Sq=4096, Skv=4096)
    out_big = torch.empty(1, 24, 4096, 128, device="cuda", dtype=torch.bfloat16)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            umfa_torch.attention_forward(q, k, v, out=out_a)
            umfa_torch.attention_forward(q, k, v, mask=mask, out=out_m)
            # a far larger item count than anything this stream has seen: the one-wave-per-SIMD kernel would have to grow its partials and
            # the fp16 image of V -- the call runs on the 128-row kernel instead, which needs neither
            umfa_torch.attention_forward(*big, out=out_big)
            assert umfa_torch.last_kernel() == "fa_fwd16<bf16,128,pv16>", umfa_torch.last_kernel()
    # larger calls on the same stream after capture: a fresh eager pool, the graph's own is untouched
    with torch.cuda.stream(side):
        qb, kb, vb = _inputs(3, H=24, Sq=4096, Skv=4096)
        umfa_torch.attention_forward(qb, kb, vb)
        mb = torch.ones(1, 1, 4096, 4096, dtype=torch.bool, device="cuda").tril(300)
        umfa_torch.attention_forward(qb, kb, vb, mask=mb)
    side.synchronize()
    eager_big = umfa_torch.attention_forward(*big)
    for _ in range(3):
        out_a.fill_(7.0)
        out_m.fill_(7.0)
        out_big.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_a, ref_a) and torch.equal(out_m, ref_m)
        assert (out_big.float() - eager_big.float()).abs().max() <= 2.0 ** -7 * eager_big.float().abs().max()  # (two kernels, one answer in bf16)


def test_two_graphs_captured_on_one_stream_replay_concurrently_on_two_streams():
    """Two graphs captured on ONE stream (torch's default is one process-wide capture stream), whose kernels cut items
    (tickets + partials in scratch), must not share that scratch: each capture owns the pool its warm-up filled.  Replayed at the same time on two streams, with eager launches running on the
    capture stream as well, every result equals the serial one."""
    import umfa_torch
    cases = [_inputs(20 + s, H=6, Sq=4096, Skv=4096) for s in range(3)]
    serial = [umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32) for q, k, v in cases]
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    outs = [torch.empty_like(serial[i]) for i in range(3)]
    graphs = []
    for i in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            umfa_torch.attention_forward(*cases[i], out=outs[i])  # warm-up before EVERY capture: the capture takes this pool over
            cap.synchronize()
            with torch.cuda.graph(g, stream=cap):
                for _ in range(4):
                    umfa_torch.attention_forward(*cases[i], out=outs[i])
        graphs.append(g)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(8):
        for o in outs:
            o.fill_(3.0)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            graphs[0].replay()
        with torch.cuda.stream(s2):
            graphs[1].replay()
        with torch.cuda.stream(cap):  # eager launches on the capture stream, overlapping both replays
            for _ in range(4):
                umfa_torch.attention_forward(*cases[2], out=outs[2])
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(outs[i], serial[i]), (rep, i)
    del graphs
    umfa_torch.release_scratch(cap)  # the captures' private pools and the capture stream's eager pool


def test_sync_entry_restores_current_device_and_rejects_unknown_mask_type():
    import ctypes
    import umfa
    from umfa._ffi import _lib
    dev_before = torch.cuda.current_device()
    q = np.random.default_rng(0).standard_normal((1, 1, 32, 32)).astype(np.float32)
    with umfa.MFAContext() as ctx:
        umfa.flash_attention_forward(ctx, q, q, q, input_precision="fp32", intermediate_precision="fp32", layout="bhsd")
        assert torch.cuda.current_device() == dev_before
        bufs = [umfa.MFABuffer(ctx, a) for a in (q, q, q, np.zeros_like(q))]
        shp = (ctypes.c_int64 * 2)(4, 4)
        rc = _lib.mfa_attention_forward(ctx.handle, *(b.handle for b in bufs), 1, 32, 32, 1, 32, 0.2, False, 2, 2, 2, False,
                                        False, False, False, ctypes.c_void_p(q.ctypes.data), 16, shp, shp, 2, 3, 0)
        assert rc == 1  # UMFA_MASK_TYPE_WINDOW exists on the in-stream entry only
        for b in bufs:
            b.close()


def test_split_kv_forward_in_a_graph_survives_many_replays(umfa_opts):
    """fa_fwd16's split-KV fold inside a captured graph.  Its tickets used to be zeroed by a hipMemsetAsync in front of every
    launch; as a memset NODE in front of a kernel whose agent-scope atomics bypass the L2 that left some tickets non-zero on
    later replays: the last part never saw "everybody has drawn", nobody folded, and O kept whatever the buffer held (the old
    result -- invisible unless the outputs are cleared between replays, which is what this test does).  The tickets are now
    zeroed once per block and reset by the folding workgroup: no memset node in the graph."""
    import umfa_torch
    umfa_opts(force_w64=0)  # (this file forces the w64 kernel for its other tests)
    torch.manual_seed(8)
    q, k, v = (torch.randn(1, 2, 512, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))

    def fn():
        o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
        return o, lse, torch.matmul(q.float(), k.float().transpose(-1, -2))
    eager = [t.clone() for t in fn()]
    assert fam(umfa_torch.last_kernel()) == "fa_fwd16<bf16,128>"  # 8 items on 256 CUs: split into key ranges
    torch.cuda.synchronize()
    side, replay_on = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            cap = fn()
    torch.cuda.synchronize()
    for rep in range(8):
        for t in cap:
            t.zero_()
        replay_on.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(replay_on):
            g.replay()
        torch.cuda.synchronize()
        for a, b in zip(eager, cap):
            assert torch.equal(a, b), rep


def test_capture_pool_goes_away_with_its_graph():
    """A capture-private scratch pool is tied to the graph it was captured into (a HIP user object on the graph; the next eager
    call frees pools whose graphs are gone).  Before: one pool per capture stayed allocated for the life of the context -- a
    process that keeps re-capturing leaked ~110 MB per FLUX-shape capture."""
    import gc
    import umfa_torch
    q, k, v = (torch.randn(1, 24, 4096, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out = torch.empty_like(q)
    side = torch.cuda.Stream()
    used = []
    for it in range(24):
        with torch.cuda.stream(side):
            umfa_torch.attention_forward(q, k, v, out=out)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                umfa_torch.attention_forward(q, k, v, out=out)
        g.replay()
        torch.cuda.synchronize()
        del g
        gc.collect()
        umfa_torch.attention_forward(q, k, v, out=out)  # an eager call: the reaping point
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used.append(total - free)
    assert used[-1] - used[4] < 64 << 20, [u >> 20 for u in used]  # 19 leaked FLUX pools would be ~2 GB


def test_two_shapes_in_one_capture_after_warming_both(umfa_opts):
    """[A, B] captured after warming A then B, where B needs MORE ticket words and SMALLER partials than A: moving the ticket
    area used to take a fresh block sized for B alone, and the capture then failed at A with error 2 (found by the fuzz's graph
    leg).  The partial area of a ticketed block never shrinks now."""
    import umfa_torch
    umfa_opts(force_w64=0)
    torch.manual_seed(3)
    a = [torch.randn(1, 6, 1024, 128, device="cuda", dtype=torch.float16) for _ in range(3)]
    b = [torch.randn(1, 6, 2048, 64, device="cuda", dtype=torch.float16) for _ in range(3)]
    ea = umfa_torch.attention_forward(*a, out_dtype=torch.float32).clone()
    eb = umfa_torch.attention_forward(*b, out_dtype=torch.float32).clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        umfa_torch.attention_forward(*a, out_dtype=torch.float32)
        umfa_torch.attention_forward(*b, out_dtype=torch.float32)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ca = umfa_torch.attention_forward(*a, out_dtype=torch.float32)
            cb = umfa_torch.attention_forward(*b, out_dtype=torch.float32)
    for _ in range(3):
        ca.zero_()
        cb.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(ca, ea) and torch.equal(cb, eb)
