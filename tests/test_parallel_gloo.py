"""N>1 path on CPU: world_size-2 gloo processes shard (batch x head) pairs, compute their shard (the CPU oracle
stands in for the HIP kernel, which cannot run here) and all-gather O; the result must equal the unsharded one."""
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

    def fn(q, k, v):
        o = oracle.sdpa_forward(np.ascontiguousarray(q.numpy()), np.ascontiguousarray(k.numpy()),
                                np.ascontiguousarray(v.numpy()), causal=causal)
        return torch.from_numpy(o)
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
    ok = torch.equal(full, ref)
    mode = parallel.plan(shape[0], shape[1], shape[2], world)
    Path(out_dir, f"r{rank}.txt").write_text(f"{int(ok)} {mode} {tuple(full.shape)}")
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,causal,mode", [((1, 4, 48, 16), False, "heads"), ((1, 3, 40, 16), True, "heads"),
                                               ((2, 1, 33, 8), False, "pairs"), ((1, 1, 64, 8), False, "rows")])
def test_two_rank_sharding_matches_unsharded(tmp_path, shape, causal, mode):
    port = 29500 + (os.getpid() + hash(shape)) % 2000
    mp.spawn(_worker, args=(2, port, shape, causal, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
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
