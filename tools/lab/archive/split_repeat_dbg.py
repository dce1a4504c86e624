import sys
sys.path[:0] = ['/root/repo', '/root/repo/universal-metal-flash-attention_amd']
import torch, umfa_torch
torch.manual_seed(9)
B, Hq, Hkv, S, D = 1, 32, 8, 1024, 128
g = Hq // Hkv
q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16)
v = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16)
ke, ve = k.repeat_interleave(g, 1).contiguous(), v.repeat_interleave(g, 1).contiguous()
for fs in (0, 1, 2, 3):
    opts = {"no_split": 1} if fs == 1 else ({"force_split": fs} if fs else {})
    with umfa_torch.options(**opts):
        outs = [umfa_torch.attention_forward(q, ke, ve, out_dtype=torch.float32, return_lse=True) for _ in range(4)]
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
        qv = q.view(B * Hkv, g, S, D)
        kv = k.view(B * Hkv, 1, S, D).expand(B * Hkv, g, S, D)
        vv = v.view(B * Hkv, 1, S, D).expand(B * Hkv, g, S, D)
        og, lg = umfa_torch.attention_forward(qv, kv, vv, out_dtype=torch.float32, return_lse=True)
        print("force", fs, "repeatable", same, "gqa view == expanded", torch.equal(og.view(B, Hq, S, D), outs[0][0]), torch.equal(lg, outs[0][1]), umfa_torch.last_kernel(),
              float((og.view(B, Hq, S, D) - outs[0][0]).abs().max()))
