import sys
sys.path[:0] = ['/root/repo', '/root/repo/universal-metal-flash-attention_amd', '/root/repo/tools']
import torch, umfa_torch
torch.manual_seed(0)
dt = torch.float16
for D in (80, 96, 40, 128, 72):
    for Sq, Skv in ((1, 130), (8, 128), (200, 130), (64, 256)):
        for mk in ("pad", "rand2d", "none_noflags"):
            B, H = 2, 2
            q = torch.randn(B, H, Sq, D, device="cuda", dtype=dt)
            k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=dt) for _ in range(2))
            if mk == "pad":
                keep = (torch.arange(Skv, device="cuda") < Skv - 7)[None, None, None, :]
            else:
                keep = (torch.rand(1, 1, Sq, Skv, device="cuda") < 0.7); keep[..., 0] = True
            opts = {"no_mask_flags": 1} if mk == "none_noflags" else {}
            with umfa_torch.options(**opts):
                o = umfa_torch.attention_forward(q, k, v, mask=keep, out_dtype=torch.float32)
            kern = umfa_torch.last_kernel()
            ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=keep)
            e = float((o - ref).abs().max())
            print(D, Sq, Skv, mk, kern, 'err %.2e' % e, "BAD" if e > 1e-3 else "")
