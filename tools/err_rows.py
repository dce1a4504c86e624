#!/usr/bin/env python3
"""Which rows carry the excess error of the deferred-max kernel?  Per-row max error (w64, in-tree) vs row statistics."""
import sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import numpy as np, torch
from ab_inproc import Lib
from oracle import oracle, parity
L = Lib("intree", "intree")
L0 = Lib("tau0", "tools/lab_bin/libMFAFFI_tau0.so")
B, H, S, D = 1, 64, 1024, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
rows = np.arange(0, S, 1)
qb, kb, vb = parity.bits(q), parity.bits(k), parity.bits(v)
ref = oracle.sdpa_forward_rows(qb, kb, vb, rows).astype(np.float64)
o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
L.forward(q, k, v, o, False); torch.cuda.synchronize()
got = o.cpu().numpy()[:, :, rows].astype(np.float64)
err = np.abs(got - ref).max(-1)[0]            # [H, rows]
L0.forward(q, k, v, o, False); torch.cuda.synchronize()
err0 = np.abs(o.cpu().numpy()[:, :, rows].astype(np.float64) - ref).max(-1)[0]
scale = np.abs(ref).max()
qf = oracle.to_f32(qb).astype(np.float64)[0][:, rows]; kf = oracle.to_f32(kb).astype(np.float64)[0]
s = np.einsum("hid,hjd->hij", qf, kf) * (D ** -0.5) * 1.4426950408889634   # log2 units
tmax = s.reshape(H, len(rows), S // 64, 64).max(-1)      # per-tile row max
run = np.maximum.accumulate(tmax, axis=-1)
gap0 = run[..., -1] - tmax[..., 0]                      # global max over tile-0 max
# emulate the wave-uniform deferred reference: waves = 64 consecutive rows
ref_m = np.empty_like(tmax)
for h in range(H):
    for w0 in range(0, len(rows), 64):
        m = np.full(64, -np.inf)
        for t in range(S // 64):
            mc = tmax[h, w0:w0 + 64, t]
            if np.any(mc - m > 6.0):
                m = np.maximum(m, mc)
            ref_m[h, w0:w0 + 64, t] = m
over = (tmax - ref_m).max(-1)                            # largest exponent of any P of the row
idx = np.dstack(np.unravel_index(np.argsort(-err, axis=None)[:12], err.shape))[0]
print("scale", scale, "median row err/scale", np.median(err) / scale)
for h, r in idx:
    print(f"h{h} row{rows[r]} err/scale {err[h, r] / scale:.2e} tau0 {err0[h, r] / scale:.2e} gap0 {gap0[h, r]:.2f} max-exponent-over-ref {over[h, r]:.2f} lse2 {np.log2(np.exp2(s[h, r] - s[h, r].max()).sum()):.2f}")
print("corr(err, over)", np.corrcoef(err.ravel(), over.ravel())[0, 1], " mean err by over-bin:")
for lo in range(0, 7):
    sel = (over >= lo) & (over < lo + 1)
    if sel.any():
        print(lo, int(sel.sum()), float(err[sel].mean() / scale), float(err[sel].max() / scale))
