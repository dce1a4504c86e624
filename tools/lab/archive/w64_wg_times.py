#!/usr/bin/env python3
"""Lab (-DW64_LAB_STAMPS build): per-workgroup start / end of one launch, grouped by XCD and by the workgroup's role in the
stream-K schedule -- where does the launch's tail come from?   python tools/lab/w64_wg_times.py B H S D [lib [blockdiag<docs> | window<half width>]]"""
import ctypes
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
if len(sys.argv) > 5:
    os.environ["UMFA_LIBRARY"] = str(Path(sys.argv[5]).resolve())
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import numpy as np
import torch
import umfa_torch
B, H, S, D = (int(x) for x in sys.argv[1:5])
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
mask = None
if len(sys.argv) > 6 and sys.argv[6].startswith("blockdiag"):
    i_ = torch.arange(S, device="cuda") // (S // int(sys.argv[6][9:] or 4))
    mask = (i_[:, None] == i_[None, :])[None, None].contiguous()
window = None
if len(sys.argv) > 6 and sys.argv[6].startswith("window"):
    window = (int(sys.argv[6][6:] or 512),) * 2
for rep in range(6):
    o, lse = umfa_torch.attention_forward(q, k, v, return_lse=True, out=out, mask=mask, window=window)
torch.cuda.synchronize()
G = 256
raw = lse.cpu().numpy().view(np.uint64)[: G * 8].reshape(G, 8)
ok = (raw[:, 1] > raw[:, 0]) & (raw[:, 1] - raw[:, 0] < 10**8)
t0 = raw[ok, 0].min()
start = (raw[:, 0].astype(np.int64) - int(t0)) / 100.0
end = (raw[:, 1].astype(np.int64) - int(t0)) / 100.0
clk = (raw[:, 3] - raw[:, 2]).astype(np.float64) / np.maximum((raw[:, 1] - raw[:, 0]).astype(np.float64) / 100.0, 1e-9)
print(umfa_torch.last_kernel(), "valid", int(ok.sum()), "span", round(float(end[ok].max()), 1))
for x in range(8):
    m = ok & (np.arange(G) % 8 == x)
    print(f"xcd {x}: n {int(m.sum()):3d} end med {np.median(end[m]):7.1f} max {end[m].max():7.1f} min {end[m].min():7.1f}  dur med {np.median(end[m]-start[m]):7.1f}  clock med {np.median(clk[m]):6.0f} MHz")
# xcd_remap(blockIdx, G): w = base(x) + (id >> 3); slices: w even / odd parts of the 128 shared items at FLUX
def remap(i, n):
    qn, r, x = n >> 3, n & 7, i & 7
    base = x * (qn + 1) if x < r else r * (qn + 1) + (x - r) * qn
    return base + (i >> 3)
w = np.array([remap(i, G) for i in range(G)])
for name, m in (("w even (starts a shared item: folder)", w % 2 == 0), ("w odd (later part: publisher)", w % 2 == 1)):
    m = m & ok
    print(f"{name}: end med {np.median(end[m]):7.1f} max {end[m].max():7.1f}")
order = np.argsort(end)
print("last 12 to finish (blockIdx, xcd, w, end):", [(int(i), int(i % 8), int(w[i]), round(float(end[i]), 1)) for i in order[-12:] if ok[i]])
print("first 6 to finish:", [(int(i), int(i % 8), int(w[i]), round(float(end[i]), 1)) for i in order[:6] if ok[i]])
acct = raw[:, 4:8].astype(np.float64).sum(1) / np.maximum(clk, 1e-9)
print("per workgroup, outside the four stamped phases (kernel entry -> first segment, per-segment set-up, exit): med %.1f us, min %.1f, max %.1f" % (
    float(np.median((end - start - acct)[ok])), float((end - start - acct)[ok].min()), float((end - start - acct)[ok].max())))
print("pro/loop/drain/epi med us:", [round(float(np.median(raw[ok, 4 + j].astype(np.float64) / np.median(clk[ok]))), 1) for j in range(4)])
