#!/usr/bin/env python3
"""Lab: per-workgroup phase stamps (W64_LAB_STAMPS build via UMFA_LIBRARY) of the mask-tensor kernel against the unmasked one."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import numpy as np
import torch
import umfa_torch
B, H, S, D = 4, 16, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
i = torch.arange(S, device="cuda")
cases = {"none": None, "all_open [1,1,S,S]": torch.ones(1, 1, S, S, dtype=torch.bool, device="cuda"),
         "padding 73 % [B,1,1,S]": (i < int(S * 0.73))[None, None, None, :].expand(B, 1, 1, S).contiguous()}
for name, m in cases.items():
    for _ in range(3):
        o, lse = umfa_torch.attention_forward(q, k, v, mask=m, return_lse=True)
    torch.cuda.synchronize()
    raw = lse.cpu().numpy().view(np.uint64)[: 256 * 8].reshape(256, 8)
    ok = (raw[:, 1] > raw[:, 0]) & (raw[:, 1] - raw[:, 0] < 10**8) & (raw[:, 3] > raw[:, 2]) & (raw[:, 3] - raw[:, 2] < 10**10)
    raw = raw[ok]
    rt = (raw[:, 1] - raw[:, 0]).astype(np.float64) / 100.0
    ck = (raw[:, 3] - raw[:, 2]).astype(np.float64)
    clk = np.median(ck / rt)
    seg = raw[:, 4:8].astype(np.float64) / clk
    print(f"{name:28s} {umfa_torch.last_kernel():36s} valid {int(ok.sum())}/256  per-WG us med {np.median(rt):7.1f} max {rt.max():7.1f}  clock {clk:5.0f} MHz | "
          f"prologue {np.median(seg[:, 0]):6.1f} loop {np.median(seg[:, 1]):7.1f} drain {np.median(seg[:, 2]):5.1f} epilogue {np.median(seg[:, 3]):6.1f}")
