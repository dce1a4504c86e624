import sys, random
sys.path[:0] = ['/root/repo', '/root/repo/universal-metal-flash-attention_amd', '/root/repo/tools/lab']
import torch, umfa_torch
import value_fuzz as vf
seed = 45
rng = random.Random(seed + 9700000)
dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16]); D = rng.choice([64, 128]); B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 5])
nqb = rng.choice([2, 2, 4, 4, 6, 8, 10, 16]); Sq = 128 * nqb - rng.choice([0, 0, 0, 1, 17, 64, 127])
Skv = rng.choice([Sq, Sq, Sq, Sq + 64, Sq + 1000, max(Sq - 100, 1), max(Sq // 2, 1), 65, 2 * Sq]); strided = rng.random() < 0.25
g = torch.Generator(device="cuda").manual_seed(seed)
mk = lambda S: torch.randn(B, H, S, D, device="cuda", generator=g).to(dt)
q, k, v = mk(Sq), mk(Skv), mk(Skv)
kind = rng.choice(["plain", "plain"] + vf.KINDS)
q, k, v = vf.transform(rng, q, k, v, kind)
vreg = rng.choice(["plain", "plain", "plain", "outlier", "tiny", "row_scaled"])
v = v.clone(); idx = (rng.randrange(B), rng.randrange(H), rng.randrange(Skv), rng.randrange(D)); val = rng.choice([3.0e8, -7.0e9, 70000.0]); v[idx] = val
print(dt, D, B, H, Sq, Skv, kind, vreg, idx, val)
i = torch.arange(Sq, device="cuda")[:, None]; j = torch.arange(Skv, device="cuda")[None, :]
for scale_q in (1.0, 0.7):
    qq = (q.float() * scale_q + (0.1 if scale_q != 1.0 else 0)).to(dt)
    ref, _ = vf.ref64(qq, k, v, D ** -0.5, j <= i)
    for cb in (1, 2):
        with umfa_torch.options(cbal=cb, no_w64=1, cbal_delta=1):
            o = umfa_torch.attention_forward(qq, k, v, causal=True, out_dtype=torch.float32)
        rel = ((o.double() - ref).abs().amax(dim=(2, 3)) / ref.abs().amax(dim=(2, 3)).clamp_min(1e-30))
        print('scale', scale_q, 'cbal', cb, rel.flatten().tolist())
qq = q
ref, _ = vf.ref64(qq, k, v, D ** -0.5, j <= i)
for cb in (1, 2):
    with umfa_torch.options(cbal=cb, no_w64=1, cbal_delta=1):
        o = umfa_torch.attention_forward(qq, k, v, causal=True, out_dtype=torch.float32)
    err = (o.double() - ref)[0, 1].abs()
    rows = err.amax(dim=1)
    top = torch.topk(rows, 6)
    print('cbal', cb, 'worst rows', top.indices.tolist(), [float(x) for x in top.values], 'col', [int(err[r].argmax()) for r in top.indices])
    r = int(top.indices[0]); c = int(err[r].argmax())
    print('  o', float(o[0, 1, r, c]), 'ref', float(ref[0, 1, r, c]), 'ratio', float(o[0,1,r,c]) / float(ref[0,1,r,c]))
    s = (qq[0,1,r].double() @ k[0,1].double().T) * D ** -0.5
    s[r+1:] = float('-inf')
    pr = torch.softmax(s, dim=0)
    print('  P(141)', float(pr[141]), 'top keys', torch.topk(pr, 3))
