import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "universal-metal-flash-attention_amd"))
os.environ["UMFA_FORCE_W64"] = "1"
import torch, numpy as np
import umfa_torch
from umfa_torch import ops
def tables(S, D):
    g = torch.Generator().manual_seed(5)
    ang = torch.rand(S, D // 2, generator=g) * 6.283
    return ang.cos().repeat_interleave(2, -1).cuda(), ang.sin().repeat_interleave(2, -1).cuda()
for dt in (torch.bfloat16, torch.float16):
    for (B, H, S, causal) in [(1, 4, 512, False), (1, 4, 768, False), (1, 4, 1024, False), (1, 4, 1280, False), (2, 2, 768, True), (2, 2, 1024, True), (1, 4, 1280, True)]:
        torch.manual_seed(11)
        q, k, v = (torch.randn(B, H, S, 128, device="cuda", dtype=dt) for _ in range(3))
        cos, sin = tables(S, 128)
        o = ops.rope_attention_forward(q, k, v, cos, sin, causal=causal)
        name = umfa_torch.last_kernel()
        r = ops.attention_forward(ops.rope_rotate(q, cos, sin), ops.rope_rotate(k, cos, sin), v, causal=causal)
        d = (o.view(torch.int16) != r.view(torch.int16))
        idx = d.nonzero()
        print(dt, (B, H, S, causal), name, "mismatch", int(d.sum()), "rows", sorted(set(idx[:, 2].tolist()))[:12], "max|d|", float((o.float() - r.float()).abs().max()), flush=True)
        # tiny-q check: how many fp16-subnormal rotated values are there?
        if dt == torch.float16:
            qr = ops.rope_rotate(q, cos, sin)
            print("   subnormal rotated q:", int(((qr.float().abs() < 6.1e-5) & (qr != 0)).sum()), "zeros:", int((qr == 0).sum()))
