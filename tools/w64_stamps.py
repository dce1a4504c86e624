#!/usr/bin/env python3
"""Lab: read the per-workgroup clock stamps a -DW64_LAB_STAMPS build leaves in the LSE buffer."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import numpy as np
import torch
import umfa_torch
B, H, S, D = (int(x) for x in sys.argv[1:5])
MODE = sys.argv[5] if len(sys.argv) > 5 else "bf16"   # bf16 | blockwise | blockwise_fp8pv
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for it in range(4):
    if MODE == "bf16":
        o, lse = umfa_torch.attention_forward(q, k, v, return_lse=True)
    else:
        o, lse = umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode=MODE, return_lse=True)
torch.cuda.synchronize()
raw = lse.cpu().numpy().view(np.uint64)[: 256 * 8].reshape(256, 8)
ok = (raw[:, 1] > raw[:, 0]) & (raw[:, 1] - raw[:, 0] < 10**8) & (raw[:, 3] > raw[:, 2]) & (raw[:, 3] - raw[:, 2] < 10**10)
raw = raw[ok]  # stamps of workgroups whose slot a later LSE store overwrote are dropped
rt = (raw[:, 1] - raw[:, 0]).astype(np.float64) / 100.0  # us
ck = (raw[:, 3] - raw[:, 2]).astype(np.float64)
span = (raw[:, 1].max() - raw[:, 0].min()) / 100.0
clk = np.median(ck / rt)  # MHz = cycles per us
seg = raw[:, 4:8].astype(np.float64) / clk
print("   per-WG us (median): prologue %.1f  main loop %.1f  drain %.1f  epilogue+fold %.1f   (max epilogue %.1f)" % (
    np.median(seg[:, 0]), np.median(seg[:, 1]), np.median(seg[:, 2]), np.median(seg[:, 3]), seg[:, 3].max()))
print(f"{umfa_torch.last_kernel()} valid {int(ok.sum())}/256 per-WG time us: min {rt.min():.1f} med {np.median(rt):.1f} max {rt.max():.1f}; span {span:.1f} us; "
      f"clock MHz: min {(ck/rt).min():.0f} med {np.median(ck/rt):.0f} max {(ck/rt).max():.0f}; start skew {(raw[:,0].max()-raw[:,0].min())/100.0:.1f} us")
if len(sys.argv) > 6 and sys.argv[6] == "fs":
    # W64_LAB_FSTAMP build: dbg[4..7] = 8 x uint32 accumulated cycle deltas; argv[7] = comma list of the stamped gaps, argv[8] = steady tiles per WG
    gaps = [int(x) for x in sys.argv[7].split(",")]
    acc = raw[:, 4:8].copy().view(np.uint32).reshape(len(raw), 8).astype(np.float64)
    tot = acc[:, :len(gaps)].sum(1)
    med = np.median(acc, 0)
    names = [f"{gaps[-1]}->next {gaps[0]}"] + [f"{gaps[i-1]}->{gaps[i]}" for i in range(1, len(gaps))]
    share = med[:len(gaps)] / med[:len(gaps)].sum()
    print("   fine stamps (median over WGs, share of the steady tile):")
    for n, m, sh in zip(names, med, share):
        print(f"      {n:>14}: {sh * 100:5.1f} %")
