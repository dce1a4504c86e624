"""Soak seed 820404 of the all-legs fuzz (run_decode_case: B5 H3 Sq4 Skv255 D128 bf16, force_split 2, decode form): 'not bitwise repeatable' once
in 1000 x 22 legs.  How often, how far apart, and with which options."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(Path(__file__).resolve().parent)]
import umfa_torch  # noqa: E402
import value_fuzz as f  # noqa: E402

fails = sum(1 for _ in range(300) if f.run_decode_case(820404))
print("seed 820404 through the leg, 300 times: failures", fails)
g = torch.Generator(device="cuda").manual_seed(820404)
B, H, Sq, Skv, D = 5, 3, 4, 255, 128
q = torch.randn(B, H, Sq, D, device="cuda", generator=g).to(torch.bfloat16)
k = torch.randn(B, H, Skv, D, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B, H, Skv, D, device="cuda", generator=g).to(torch.bfloat16)
for opts in ({"decode_ks": 0, "force_split": 2}, {"decode_ks": 2, "force_split": 2}, {"decode_ks": 0}, {"decode_ks": 0, "force_split": 3}, {"decode_ks": 1, "force_split": 2}):
    with umfa_torch.options(**opts):
        ref = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
        torch.cuda.synchronize()
        kern = umfa_torch.last_kernel()
        diff, worst = 0, 0.0
        for _ in range(3000):
            o = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32)
            if not torch.equal(o, ref):
                diff += 1
                worst = max(worst, float((o - ref).abs().max() / ref.abs().max()))
    print(opts, kern, "3000 launches: differing", diff, "worst rel", worst, flush=True)

# ... and in bulk, the comparison left on the device (no host round trip between launches)
for (Bx, Hx, Sqx, Skvx, fs) in ((5, 3, 4, 255, 2), (5, 3, 1, 255, 2), (2, 8, 4, 129, 2), (5, 3, 4, 1000, 7), (1, 8, 8, 4097, 16)):
    qx = torch.randn(Bx, Hx, Sqx, D, device="cuda", generator=g).to(torch.bfloat16)
    kx = torch.randn(Bx, Hx, Skvx, D, device="cuda", generator=g).to(torch.bfloat16)
    vx = torch.randn(Bx, Hx, Skvx, D, device="cuda", generator=g).to(torch.bfloat16)
    with umfa_torch.options(decode_ks=0, force_split=fs):
        ref, rlse = umfa_torch.attention_forward(qx, kx, vx, out_dtype=torch.float32, return_lse=True)
        kern = umfa_torch.last_kernel()
        mism = torch.zeros((), device="cuda", dtype=torch.int64)
        o = torch.empty_like(ref)
        n = 40000
        for i in range(n):
            if i & 1:
                o = umfa_torch.attention_forward(qx, kx, vx, out_dtype=torch.float32)
            else:
                o, _ = umfa_torch.attention_forward(qx, kx, vx, out_dtype=torch.float32, return_lse=True)
            mism += (o != ref).any()
        torch.cuda.synchronize()
    print((Bx, Hx, Sqx, Skvx, fs), kern, n, "launches (LSE on every other): differing", int(mism.item()), flush=True)
