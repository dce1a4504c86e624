"""Backward at larger / ragged head_dim-128 shapes against an fp32 torch restatement on the GPU (dq2 / dq1 / dkdv pinned)."""
import sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "universal-metal-flash-attention_amd"), os.path.join(ROOT, "tests")]
import torch, umfa_torch
import test_gpu_fuzz as F
bad = 0
for i, (B, H, Sq, Skv, causal, dt) in enumerate([(1, 2, 3000, 2500, False, torch.bfloat16), (1, 2, 2500, 3000, True, torch.bfloat16), (2, 3, 4096, 4096, True, torch.bfloat16),
                                               (1, 4, 4096, 4096, False, torch.float16), (1, 1, 4160, 4100, False, torch.bfloat16), (1, 2, 1999, 2001, True, torch.float16),
                                               (1, 2, 64, 4096, False, torch.bfloat16), (1, 2, 4096, 64, False, torch.bfloat16), (1, 3, 129, 8192, True, torch.bfloat16)]):
    for force in ("0", "1", "2"):
        os.environ["UMFA_BWD_DQ"] = force
        g = torch.Generator(device="cuda").manual_seed(i)
        q, k, v = (torch.randn(B, H, s, 128, device="cuda", dtype=dt, generator=g) for s in (Sq, Skv, Skv))
        do = torch.randn(B, H, Sq, 128, device="cuda", dtype=dt, generator=g)
        qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
        F._ref(qr, kr, vr, 128 ** -0.5, causal, None).backward(do.float())
        qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        umfa_torch.scaled_dot_product_attention(qg, kg, vg, is_causal=causal).backward(do)
        errs = []
        for got, ref in ((qg.grad, qr.grad), (kg.grad, kr.grad), (vg.grad, vr.grad)):
            errs.append(((got.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-3)).item())
        ok = all(e < (3e-2 if dt == torch.bfloat16 else 8e-3) for e in errs) and all(torch.isfinite(t.grad).all() for t in (qg, kg, vg))
        bad += not ok
        print(("ok  " if ok else "FAIL"), (B, H, Sq, Skv, causal, str(dt)[6:]), "dq kernel", force, [round(e, 4) for e in errs], umfa_torch.last_kernel(), flush=True)
print("failures", bad)
