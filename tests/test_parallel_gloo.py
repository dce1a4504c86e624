"""N>1 path on CPU: world_size-2 and -4 gloo processes shard (batch x head) pairs / heads / query rows, compute their
shard (the CPU oracle stands in for the HIP kernel, which cannot run here) and all-gather O; the result must equal the
unsharded one.  Also the strong-scaling form bench.py times at N > 1 (heads dealt round-robin, per-head in-place
all-gather into the final tensor: umfa_torch.parallel.overlapped_sharded_sdpa)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _oracle_fn(causal):
    from oracle import oracle

    def fn(q, k, v, row_offset=0, out=None):
        qn, kn, vn = (np.ascontiguousarray(t.numpy()) for t in (q, k, v))
        if causal and row_offset:
            # a query-row shard of a causal problem: local row i is global row i + row_offset.  The oracle's row-subset
            # entry keeps absolute row numbers: place the shard's rows at their global positions of a full-size Q
            full_q = np.zeros((qn.shape[0], qn.shape[1], row_offset + qn.shape[2], qn.shape[3]), qn.dtype)
            full_q[:, :, row_offset:] = qn
            o = oracle.sdpa_forward_rows(full_q, kn, vn, np.arange(row_offset, row_offset + qn.shape[2]), causal=True)
        else:
            o = oracle.sdpa_forward(qn, kn, vn, causal=causal)
        o = torch.from_numpy(np.ascontiguousarray(o))
        if out is not None:
            out.copy_(o)
            return out
        return o
    return fn


def _worker(rank, world, port, shape, causal, out_dir):
    for p in (str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from umfa_torch import parallel
    torch.manual_seed(0)  # every rank holds the same full problem
    q, k, v = (torch.randn(shape) for _ in range(3))
    full = parallel.sharded_sdpa(q, k, v, causal=causal, gather=True, attention_fn=_oracle_fn(causal))
    ref = _oracle_fn(causal)(q, k, v)
    ok = torch.equal(full, ref) if not (causal and parallel.plan(*shape[:3], world) == "rows") else bool((full - ref).abs().max() < 1e-6)
    mode = parallel.plan(shape[0], shape[1], shape[2], world)
    if shape[0] == 1 and shape[1] % world == 0:  # the overlapped strong-scaling form (bench.py N > 1 headline)
        out_full = torch.full(shape, float("nan"))
        parallel.overlapped_sharded_sdpa(q, k, v, out_full, attention_fn=_oracle_fn(causal))
        ok = ok and torch.equal(out_full, ref)
        heads = sorted(h for r in range(world) for a, b, _, _ in parallel.owned_heads(shape[1], world, r) for h in range(a, b))
        ok = ok and heads == list(range(shape[1]))  # every head owned exactly once
    Path(out_dir, f"r{rank}.txt").write_text(f"{int(ok)} {mode} {tuple(full.shape)}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,causal,mode", [(2, (1, 4, 48, 16), False, "heads"), (2, (1, 3, 40, 16), True, "heads"),
                                                     (2, (2, 1, 33, 8), False, "pairs"), (2, (1, 1, 64, 8), False, "rows"),
                                                     (2, (1, 1, 64, 8), True, "rows"), (4, (1, 8, 32, 16), False, "heads"),
                                                     (4, (2, 4, 24, 8), True, "heads"), (4, (1, 2, 48, 8), True, "rows")])
def test_sharding_matches_unsharded(tmp_path, world, shape, causal, mode):
    port = 29500 + (os.getpid() + hash((world, shape, causal))) % 2000
    mp.spawn(_worker, args=(world, port, shape, causal, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, m, shp = (tmp_path / f"r{r}.txt").read_text().split(" ", 2)
        assert ok == "1" and m == mode, (r, ok, m, shp)


def test_shard_ranges_cover_exactly():
    from umfa_torch import parallel
    for n in (1, 3, 24, 32, 7):
        for world in (1, 2, 4, 8):
            spans = [parallel.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.plan(1, 24, 4096, 8) == "heads" and parallel.plan(1, 32, 32768, 8) == "heads"  # configs 3, 5
    assert parallel.plan(4, 1, 128, 4) == "pairs" and parallel.plan(1, 1, 128, 2) == "rows"
