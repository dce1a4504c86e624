"""Lab: phase times of the pinned bwd16_dkdv tile (a -DBWD16_LAB_STAMP build leaves them in the head of the LSE buffer).
UMFA_LIBRARY=tools/lab_bin/libMFAFFI_bwdstamp.so python tools/lab/bwd_stamps.py"""
import sys, os
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "universal-metal-flash-attention_amd")]
import numpy as np, torch
import umfa_torch
B, H, S, D = 1, 24, 4096, 128
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
o, lse = umfa_torch.attention_forward(q, k, v, return_lse=True)
for _ in range(3):
    l2 = lse.clone()
    umfa_torch.attention_backward(do, q, k, v, o, l2, scale=D ** -0.5)
torch.cuda.synchronize()
full = l2.cpu().numpy().view(np.uint32)[: 768 * 16].reshape(768, 16)
raw = full[:, :6].astype(np.float64)
tot = raw.sum(1)
names = ["stage issue", "P1a (S, dP of sub-tile 0)", "P1b (+ softmax 0)", "P2a (+ softmax 1)", "P2b", "vmcnt + barrier"]
med = np.median(raw, 0)
print("tiles per workgroup 64; median per-tile s_memtime ticks per phase (share):")
for n, m in zip(names, med):
    print(f"  {n:28s} {m / 64:8.1f}  {m / med.sum() * 100:5.1f} %")
print("  total per tile", med.sum() / 64)

pro, loop, epi = (full[:, 6].astype(np.float64) / 100, full[:, 7].astype(np.float64) / 100, full[:, 8].astype(np.float64) / 100)
print(f"per workgroup (us, median / max): prologue {np.median(pro):.1f} / {pro.max():.1f}   tile loops {np.median(loop):.1f} / {loop.max():.1f}   epilogue {np.median(epi):.1f} / {epi.max():.1f}")
ent, ext = full[:, 9].astype(np.int64), full[:, 10].astype(np.int64)
t0 = ent.min()
span = ((ext - t0) & 0xffffffff).max() / 100
busy = ((ext - ent) & 0xffffffff).sum() / 100
print(f"kernel span {span:.1f} us; sum of workgroup lifetimes {busy:.0f} us = {busy / 256:.1f} us per CU = {busy / 256 / span * 100:.0f} % of the span")
