"""Multi-GPU sharding of SDPA over one node (one process per GPU, torch.distributed; backend "nccl" = RCCL).

The reference has no distributed path at all (one Metal queue, MFABridge.swift:245-251); this is new
design for MI355X nodes.  Attention is independent per (batch, head), so the path shards with NO data-path exchange:
rank r owns a set of heads (or, when H < world, of flattened (batch, head) pairs, then of query rows).  Only a caller
that wants the full output on every rank pays an all-gather of O, which on xGMI (point-to-point links) is a direct
exchange of equal shards:

* `sharded_sdpa`            contiguous shards, one `all_gather_into_tensor` behind the kernel (equal shards land in the
                            final tensor through a view: no pad, no cat; ragged shards are padded to the largest);
* `overlapped_sharded_sdpa` the strong-scaling form bench.py times at N > 1: heads dealt round-robin (rank r owns heads
                            r, r + N, ...), one kernel launch per owned head writing straight into its slot of the final
                            [B, H, S, D] buffer, and an IN-PLACE all-gather of head group c on a side stream while head
                            c + 1 is being computed -- the N slots of a group are adjacent, so each gather's output is a
                            contiguous slice of the final tensor and nothing is ever copied.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n units; the first n % world ranks take one extra."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def plan(B: int, H: int, Sq: int, world: int) -> str:
    """Which axis to shard: heads, else (batch x head) pairs, else query rows."""
    if H >= world:
        return "heads"
    if B * H >= world:
        return "pairs"
    return "rows"


def local_slices(q, k, v, world: int, rank: int):
    """Views of the FULL [B,H,S,D] tensors that this rank computes (no copies for heads/rows)."""
    B, H, Sq, _ = q.shape
    mode = plan(B, H, Sq, world)
    if mode == "heads":
        a, b = shard_range(H, world, rank)
        return mode, q[:, a:b], k[:, a:b], v[:, a:b]
    if mode == "pairs":
        a, b = shard_range(B * H, world, rank)
        flat = lambda t: t.reshape(1, B * H, t.shape[2], t.shape[3])  # noqa: E731
        return mode, flat(q)[:, a:b], flat(k)[:, a:b], flat(v)[:, a:b]
    a, b = shard_range(Sq, world, rank)
    return mode, q[:, :, a:b], k, v  # every rank needs all keys of its heads, which it has


def all_gather_output(o_local: torch.Tensor, mode: str, full_shape, group=None) -> torch.Tensor:
    """Assemble the full O on every rank from per-rank shards.

    Equal shards (every BASELINE config: 24 or 32 heads over 1 / 2 / 4 / 8 ranks): ONE `all_gather_into_tensor` into
    `[world, *shard]`; for B = 1 head shards (and B*H = 1 row shards) that buffer IS the final tensor (returned as a view),
    otherwise one permuting copy puts the rank axis in place.  Ragged shards are padded to the largest and trimmed."""
    world = dist.get_world_size(group)
    B, H, Sq, D = full_shape
    axis = {"heads": 1, "pairs": 1, "rows": 2}[mode]
    n = {"heads": H, "pairs": B * H, "rows": Sq}[mode]
    sizes = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]
    if min(sizes) == max(sizes):
        buf = o_local.new_empty((world,) + tuple(o_local.shape))
        if o_local.is_cuda and dist.get_backend(group) == "gloo":  # one-device rehearsal: gloo gathers on the host
            host = torch.empty(buf.shape, dtype=buf.dtype)
            dist.all_gather_into_tensor(host.view(-1), o_local.contiguous().view(-1).cpu(), group=group)
            buf.copy_(host)
        else:
            dist.all_gather_into_tensor(buf.view(-1), o_local.contiguous().view(-1), group=group)  # flat: the form every backend takes
        full = buf.movedim(0, axis)  # [.., world, shard, ..]: adjacent to the sharded axis
        shp = list(o_local.shape)
        shp[axis] *= world
        full = full.reshape(shp)  # a view when every axis in front of the sharded one has extent 1, else one copy
        return full.reshape(B, H, Sq, D) if mode == "pairs" else full
    mx = max(sizes)
    pad_shape = list(o_local.shape)
    pad_shape[axis] = mx
    padded = o_local.new_zeros(pad_shape)
    padded.narrow(axis, 0, o_local.shape[axis]).copy_(o_local)
    buf = o_local.new_empty((world,) + tuple(pad_shape))
    dist.all_gather_into_tensor(buf.view(-1), padded.view(-1), group=group)
    full = torch.cat([buf[r].narrow(axis, 0, s) for r, s in enumerate(sizes) if s > 0], dim=axis)
    return full.reshape(B, H, Sq, D) if mode == "pairs" else full


def _default_attention(causal: bool, scale: Optional[float]):
    from .ops import attention_forward

    def fn(a, b, c, row_offset: int = 0, out=None):
        if causal and row_offset:
            # causal + query-row shard: local row i is global row i + row_offset; "key <= global row" is the sliding
            # window (left = everything, right = row_offset) of the in-stream entry: no mask tensor, tile early-exit
            return attention_forward(a, b, c, scale=scale, window=(int(b.shape[2]) + int(row_offset), int(row_offset)), out=out)
        return attention_forward(a, b, c, causal=causal, scale=scale, out=out)
    return fn


def sharded_sdpa(q, k, v, *, causal: bool = False, scale: Optional[float] = None, gather: bool = True, group=None,
                 attention_fn: Optional[Callable] = None):
    """q, k, v: the full [B,H,S,D] problem, replicated (or addressable) on every rank.  Each rank computes its
    shard with `attention_fn(q, k, v[, row_offset=])` (default: the HIP forward) and, if `gather`, all-gathers O.
    Query-row shards of a causal problem pass their first global row as `row_offset`."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if attention_fn is None:
        attention_fn = _default_attention(causal, scale)
    mode, ql, kl, vl = local_slices(q, k, v, world, rank)
    if mode == "rows" and causal:
        a, _ = shard_range(q.shape[2], world, rank)
        o_local = attention_fn(ql, kl, vl, row_offset=a)
    else:
        o_local = attention_fn(ql, kl, vl)
    if not gather or world == 1:
        return o_local
    return all_gather_output(o_local, mode, tuple(q.shape), group)


def chunk_plan(H: int, world: int, chunks=None):
    """Head chunks of the overlapped form: per-rank chunk sizes (sum = H / world).  Default: two chunks, the larger first,
    so the first chunk's all-gather runs under the second chunk's kernel and only the last (smaller) gather is exposed."""
    per = H // world
    if chunks is None:
        chunks = [per] if per < 2 else [per - per // 2, per // 2]
    if sum(chunks) != per or min(chunks) <= 0:
        raise ValueError(f"chunks {chunks} must be positive and sum to H / world = {per}")
    return list(chunks)


def owned_heads(H: int, world: int, rank: int, chunks=None):
    """Head ownership of the overlapped form: chunk c covers the contiguous heads [base_c, base_c + m_c * world) and rank
    r owns [base_c + r * m_c, base_c + (r + 1) * m_c) of it -- contiguous per rank inside a contiguous group, which is
    what lets every all-gather run in place on a slice of the final tensor."""
    out, base = [], 0
    for m in chunk_plan(H, world, chunks):
        out.append((base + rank * m, base + (rank + 1) * m, base, base + m * world))
        base += m * world
    return out


_INPLACE_OK = {}


def inplace_all_gather_ok(group, device) -> bool:
    """Does this backend's `all_gather_into_tensor` accept input = the rank's own slice of the output (NCCL / RCCL's in-place
    form) and produce the right answer?  Checked ONCE per (group, device) on 64 floats per rank, agreed over the ranks (MIN),
    so that a runtime that rejects or mishandles the aliasing costs a staging copy, not the run."""
    key = (id(group), str(device))
    if key in _INPLACE_OK:
        return _INPLACE_OK[key]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ok = dist.get_backend(group) != "gloo"
    if ok:
        try:
            buf = torch.full((world * 64,), -1.0, device=device, dtype=torch.float32)
            mine = buf[rank * 64:(rank + 1) * 64]
            mine.fill_(float(rank + 1))
            dist.all_gather_into_tensor(buf, mine, group=group)
            expect = torch.arange(1, world + 1, device=device, dtype=torch.float32).repeat_interleave(64)
            ok = bool(torch.equal(buf, expect))
        except Exception:  # noqa: BLE001  (argument validation of a torch build that refuses aliased tensors)
            ok = False
        flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = bool(int(flag.item()) == 1)
    _INPLACE_OK[key] = ok
    return ok


def overlapped_sharded_sdpa(q, k, v, out_full: torch.Tensor, *, attention_fn: Callable, group=None, comm_stream=None,
                            chunks=None):
    """ONE [B=1, H, S, D] problem over the ranks of `group`, full O on every rank, the all-gather hidden behind compute.

    `attention_fn(qc, kc, vc, out=slot)` computes a chunk of heads ([1, m, S, D] views) into `slot`, a view of `out_full`.
    Per chunk (owned_heads): the rank's kernel writes its m heads straight into the final buffer, then the chunk's head
    group is completed by an IN-PLACE all-gather (every rank's input is its own slice of the output: no staging copy)
    issued on `comm_stream` behind an event of the compute stream -- chunk c travels over xGMI while chunk c + 1 is
    computed.  Returns `out_full`; the caller's current stream waits for the last gather before returning."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, H, S, D = q.shape
    if B != 1 or H % world != 0:
        raise ValueError("overlapped_sharded_sdpa: one batch element and H % world == 0 (BASELINE configs 3 and 5)")
    assert out_full.shape == q.shape and out_full.is_contiguous()
    on_gpu = out_full.is_cuda
    if on_gpu and comm_stream is None and world > 1:
        comm_stream = torch.cuda.Stream(device=out_full.device)
    inplace = on_gpu and world > 1 and inplace_all_gather_ok(group, out_full.device)
    for a, b, g0, g1 in owned_heads(H, world, rank, chunks):
        attention_fn(q[:, a:b], k[:, a:b], v[:, a:b], out=out_full[:, a:b])
        if world == 1:
            continue
        grp = out_full[0, g0:g1].view(-1)   # the chunk's head group: contiguous
        mine = out_full[0, a:b].view(-1)    # this rank's slice of it: the in-place all-gather's input
        if on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(out_full.device))
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                if inplace:
                    dist.all_gather_into_tensor(grp, mine, group=group)  # RCCL: sendbuff = recvbuff + rank * count is its in-place form
                elif dist.get_backend(group) != "gloo":
                    dist.all_gather_into_tensor(grp, mine.clone(), group=group)  # in-place form refused by this runtime: one staging copy
                else:  # code-path rehearsal on one device (bench.py UMFA_BENCH_ONE_DEVICE): gloo gathers on the host
                    host = torch.empty(grp.shape, dtype=grp.dtype)
                    dist.all_gather_into_tensor(host, mine.cpu(), group=group)
                    grp.copy_(host)
        else:
            dist.all_gather_into_tensor(grp, mine.clone(), group=group)  # gloo (CPU tests) has no in-place form
    if on_gpu and world > 1:
        torch.cuda.current_stream(out_full.device).wait_stream(comm_stream)
    return out_full
