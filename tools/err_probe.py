#!/usr/bin/env python3
"""Error anatomy of the bf16 forward: max-norm and rms rel-err vs the oracle for several library builds, next to the
operand-format floor (P rounded to bf16 once, everything else fp64) ON THE SAME ROWS.
python tools/err_probe.py name=path ... (path 'intree' = the built library; UMFA_NO_W64 is read at first use per lib)"""
import ctypes, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import numpy as np, torch
from ab_inproc import Lib
from oracle import oracle, parity

libs = [Lib(*s.split("=", 1)) for s in sys.argv[1:] if "=" in s]
dt = torch.float16 if "--fp16" in sys.argv else torch.bfloat16
kind = "fp16" if dt == torch.float16 else "bf16"
out = {}
for (B, H, S, D) in [(1, 256, 256, 128), (1, 128, 512, 128), (1, 64, 1024, 128), (1, 24, 4096, 128)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(dt) for _ in range(3))
    rows = parity.sample_rows(S, groups=4, width=32)
    qb, kb, vb = parity.bits(q), parity.bits(k), parity.bits(v)
    ref = oracle.sdpa_forward_rows(qb, kb, vb, rows).astype(np.float64)
    fl = np.concatenate([oracle.flash_format_floor(qb[:, h0:h0 + 8], kb[:, h0:h0 + 8], vb[:, h0:h0 + 8], rows, kind)
                         for h0 in range(0, H, 8)], axis=1).astype(np.float64)
    def stats(x):
        d = x - ref
        return {"max": float(np.abs(d).max() / np.abs(ref).max()), "rms": float(np.sqrt((d * d).mean() / (ref * ref).mean()))}
    r = {"floor": stats(fl)}
    for L in libs:
        o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        L.forward(q, k, v, o, False)
        torch.cuda.synchronize()
        r[L.name] = stats(o.cpu().numpy()[:, :, rows].astype(np.float64))
        r[L.name]["kernel"] = L.kernel()
    out[f"B{B}_H{H}_S{S}"] = r
    print(f"B{B}_H{H}_S{S}", json.dumps(r), flush=True)
