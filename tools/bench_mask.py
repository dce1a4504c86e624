#!/usr/bin/env python3
"""Masked forward timing: python tools/bench_mask.py [window] -- FLUX shape, bool sliding-window mask [1,1,S,S] and a
dense random per-head mask, with and without the tile flags (UMFA_NO_MASK_FLAGS=1)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
B, H, S, D = 1, 24, 4096, 128
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
i = torch.arange(S, device="cuda")[:, None]; j = torch.arange(S, device="cuda")[None, :]
masks = {"sliding window %d, bool [1,1,S,S]" % W: ((i - j).abs() <= W)[None, None].contiguous(),
         "key padding 3000 of 4096, bool [1,1,1,S]": (j < 3000)[None, None].contiguous(),
         "random 80 %% open, bool [1,H,S,S]": torch.rand(1, H, S, S, device="cuda") < 0.8,
         "all open, bool [1,1,S,S]": torch.ones(1, 1, S, S, dtype=torch.bool, device="cuda"),
         "additive fp32 bias N(0,1) [1,1,S,S]": torch.randn(1, 1, S, S, device="cuda"),
         "additive bf16 bias [1,H,S,S]": torch.randn(1, H, S, S, device="cuda", dtype=torch.bfloat16),
         "sliding window 512 as fp32 0 / -inf [1,1,S,S]": torch.where((i - j).abs() <= W, 0.0, float("-inf"))[None, None].contiguous()}
out = torch.empty_like(q)
def timeit(mask):
    for _ in range(3): umfa_torch.attention_forward(q, k, v, mask=mask, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); umfa_torch.attention_forward(q, k, v, mask=mask, out=out); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3
for name, m in masks.items():
    os.environ.pop("UMFA_NO_MASK_FLAGS", None)
    a = timeit(m)
    os.environ["UMFA_NO_MASK_FLAGS"] = "1"
    b = timeit(m)
    print(f"{name:45s} with tile flags {a:8.1f} us   per-score only {b:8.1f} us   [{umfa_torch.last_kernel()}]")
os.environ.pop("UMFA_NO_MASK_FLAGS", None)
a = timeit(None)
print(f"{'no mask':45s} {a:8.1f} us [{umfa_torch.last_kernel()}]")
