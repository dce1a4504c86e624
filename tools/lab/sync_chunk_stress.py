"""Stress of the chunked synchronous entries at their production floors, in a process that also runs torch: per-call wrapping (pin / unpin of
the host ranges every call), arrays of changing sizes (glibc's mmap threshold moves: some land in the heap, sharing edge pages with other
objects), torch host <-> device copies and in-stream launches in between, random chunk counts.  Every result against the one-upload form's
bounds (each inside 1e-3 of the other; a spot check against the oracle every 20th iteration).
python tools/lab/sync_chunk_stress.py ITERATIONS [SEED]"""
import random
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import umfa  # noqa: E402
import umfa_torch  # noqa: E402
from oracle import oracle  # noqa: E402

iters = int(sys.argv[1])
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nrng = np.random.default_rng(1)


def bits(shape):
    return (nrng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


junk, bad = [], 0
c0 = int(umfa_torch.get_option("sync_chunked_calls"))
with umfa.MFAContext() as ctx:
    for it in range(iters):
        B, H = rng.choice([(1, 8), (1, 12), (2, 6), (1, 24), (4, 4), (3, 5)])
        S, D = rng.choice([(1024, 128), (2048, 64), (1536, 128), (2048, 128)])
        if 5 * B * H * S * D * 2 < (17 << 20):
            S *= 2
        q, k, v = bits((B, H, S, D)), bits((B, H, S, D)), bits((B, H, S, D))
        # heap churn: arrays of many sizes come and go (the mmap threshold follows the largest freed mmap chunk)
        junk.append(np.ones(rng.choice([1 << 12, 1 << 16, 1 << 20, 3 << 20, 9 << 20, 20 << 20]), np.uint8))
        if len(junk) > 6:
            del junk[rng.randrange(len(junk))]
        t = torch.randn(rng.choice([1 << 10, 1 << 18, 1 << 22]), device="cuda")
        tc = t.cpu()                      # pageable device -> host copy by torch
        t2 = tc.to("cuda") * 2            # ... and back
        causal = rng.random() < 0.3
        chunks = rng.choice([0, 2, 3, 4, 6, 8, 16])
        kw = dict(input_precision="bf16", intermediate_precision="bf16", layout="bhsd", causal=causal)
        with umfa_torch.options(sync_chunks=1):
            o1 = umfa.flash_attention_forward(ctx, q, k, v, **kw)
        with umfa_torch.options(sync_chunks=chunks):
            oc, lc = umfa.flash_attention_forward(ctx, q, k, v, return_lse=True, **kw)
            if it % 5 == 0:
                do = bits((B, H, S, D))
                g = umfa.attention_backward(ctx, do, q, k, v, oc, lc, causal=causal, input_precision="bf16", layout="bhsd")
                with umfa_torch.options(sync_chunks=1):
                    g1 = umfa.attention_backward(ctx, do, q, k, v, oc, lc, causal=causal, input_precision="bf16", layout="bhsd")
                for a, b_ in zip(g[:3], g1[:3]):
                    if not np.isfinite(a).all() or rel(a, b_) > 2e-3:
                        bad += 1
                        print("FAIL bwd", it, (B, H, S, D), chunks, causal, rel(a, b_), flush=True)
        assert torch.allclose(t2.cpu(), tc * 2)
        r = rel(oc, o1)
        if not np.isfinite(oc).all() or r > 1e-3:
            bad += 1
            print("FAIL fwd", it, (B, H, S, D), chunks, causal, r, flush=True)
        if it % 20 == 0:
            h = rng.randrange(H)
            ref = oracle.sdpa_forward(q[:1, h:h + 1], k[:1, h:h + 1], v[:1, h:h + 1], causal=causal)
            if rel(oc[:1, h:h + 1], ref) > 1e-3:
                bad += 1
                print("FAIL oracle", it, (B, H, S, D), chunks, causal, flush=True)
print("done", iters, "failures", bad, "chunked calls", int(umfa_torch.get_option("sync_chunked_calls")) - c0)
