"""Full-size parity measurements against the CPU oracle (TEST INFRASTRUCTURE ONLY: used by tests/, by
bench.py's `parity` object and by tools/parity_probe.py -- as the checker, never as the thing measured).

rel = max|O - O_ref| / max|O_ref| over a row subset (all heads, all keys), O_ref = oracle/sdpa_ref.c (fp64) on the
already rounded inputs.  The subset is 8 groups of 32 consecutive rows spread over Sq (first and last group included),
so causal diagonals, ragged tails and every wave position of a 256-row work item are sampled.
"""
from __future__ import annotations

import numpy as np

from . import oracle


def sample_rows(Sq: int, groups: int = 8, width: int = 32) -> np.ndarray:
    if Sq <= groups * width:
        return np.arange(Sq)
    starts = np.linspace(0, Sq - width, groups).astype(np.int64)
    starts[1:-1] += 7 * np.arange(1, groups - 1)  # off the 32/64/256-row grid
    return np.unique(np.concatenate([np.arange(s, min(Sq, s + width)) for s in starts]))


def bits(t):
    """torch tensor (cpu or cuda; bf16 / fp16 / fp32) -> the numpy array the oracle takes (bf16 as uint16 bits)."""
    import torch
    t = t.detach().cpu()
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.contiguous().numpy()


def rel_err(o: np.ndarray, ref: np.ndarray) -> float:
    return float(np.abs(o.astype(np.float64) - ref.astype(np.float64)).max() / np.abs(ref).max())


def forward_rel_err(q, k, v, o, *, causal=False, scale=None, rows=None, floor_kind=None):
    """q, k, v, o: torch tensors [B,H,S,D] (o any float dtype).  Returns {"rel", "rows", ["format_floor"]}."""
    Sq = q.shape[2]
    rows = sample_rows(Sq) if rows is None else np.asarray(rows)
    qb, kb, vb = bits(q), bits(k), bits(v)
    ref = oracle.sdpa_forward_rows(qb, kb, vb, rows, scale=scale, causal=causal)
    got = o.detach().float().cpu().numpy()[:, :, rows]
    d = got.astype(np.float64) - ref.astype(np.float64)
    res = {"rel": rel_err(got, ref), "rms": float(np.sqrt((d * d).mean() / (ref.astype(np.float64) ** 2).mean())), "rows": int(rows.size)}
    if floor_kind:
        sub = rows[:: max(1, rows.size // 64)]  # the emulation holds [B,H,rows,Skv] in fp64: keep it small
        pos = np.searchsorted(rows, sub)
        fl = oracle.flash_format_floor(qb, kb, vb, sub, floor_kind, scale=scale, causal=causal)
        res["format_floor"] = rel_err(fl, ref[:, :, pos])
        df = fl.astype(np.float64) - ref[:, :, pos].astype(np.float64)
        res["format_floor_rms"] = float(np.sqrt((df * df).mean() / (ref[:, :, pos].astype(np.float64) ** 2).mean()))
    return res
