#!/usr/bin/env python3
"""GQA training step (forward + backward through torch autograd), K / V read in place vs the reference's repeat_interleave
route: B1 Hq32 Hkv8 S4096 D128 bf16, causal.  Median of event-timed iterations + peak memory."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402

B, Hq, Hkv, S, D = 1, 32, 8, 4096, 128
g = Hq // Hkv
torch.manual_seed(0)
q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
do = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16)


def step_inplace():
    o = umfa_torch.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
    o.backward(do)


def step_expand():
    o = umfa_torch.scaled_dot_product_attention(q, k.repeat_interleave(g, 1), v.repeat_interleave(g, 1), is_causal=True)
    o.backward(do)


res = {}
for name, fn in (("in_place", step_inplace), ("repeat_interleave", step_expand)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    res[name] = {"ms": round(ts[len(ts) // 2], 4), "peak_extra_MB": round((torch.cuda.max_memory_allocated() - base) / 1e6, 1), "kernel": umfa_torch.last_kernel()}
res["speedup"] = round(res["repeat_interleave"]["ms"] / res["in_place"]["ms"], 3)
print(json.dumps(res))
