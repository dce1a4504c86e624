#!/usr/bin/env python3
"""The transformers idiom -- an additive [B,1,S,S] mask 0 / torch.finfo(dtype).min, causal + padding -- on small and large launches: default against no_mask_flags
(= no tile is skipped: what the flag pass gave these masks before it classified the TERM instead of the raw bits) and against the same mask with -inf.  Graph-replayed; JSON lines."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import torch
import umfa_torch
from bench_mask_f32 import graph_us

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for (B, H, S, D) in [(2, 8, 2048, 128), (4, 16, 1024, 64), (1, 32, 4096, 128), (8, 32, 2048, 128), (4, 8, 4096, 128)]:
    for dt in (torch.bfloat16,):
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
        o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        i = torch.arange(S, device="cuda")
        lens = torch.tensor([S - (S // (4 * B)) * b for b in range(B)], device="cuda")
        keep = (i[None, :, None] >= i[None, None, :]) & (i[None, None, :] < lens[:, None, None])
        for mdt in (torch.float32, torch.bfloat16, torch.float16):
            m_min = torch.where(keep, 0.0, torch.finfo(mdt).min).to(mdt)[:, None].contiguous()
            m_inf = torch.where(keep, 0.0, float("-inf")).to(mdt)[:, None].contiguous()
            t = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m_min, out=o))
            kern = umfa_torch.last_kernel().split(" (")[0]
            with umfa_torch.options(no_mask_flags=1, no_w64_bias=1):
                t_nf = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m_min, out=o))
            t_inf = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m_inf, out=o))
            t_c = graph_us(lambda: umfa_torch.attention_forward(q, k, v, causal=True, out=o))
            rec = {"shape": f"B{B} H{H} S{S} D{D}", "mask": f"causal + padding, 0 / finfo({str(mdt).split('.')[1]}).min [B,1,S,S]", "us": round(t, 1), "kernel": kern,
                   "row128_no_tile_skipped_us": round(t_nf, 1), "same_mask_with_minus_inf_us": round(t_inf, 1), "causal_flag_no_mask_us": round(t_c, 1)}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
